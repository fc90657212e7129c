// K5 / K9: CNN forward for the four NanoCaller models (gfx950), fp32.
//
// Restates model_architect.py:36-64, model_architect_SNP_haploid.py:33-53, model_architect_indel.py:28-48,
// model_architect_indels_haploid.py:29-48: three parallel `same` convs (1x5, 5x1, 5x5)
// -> concat -> two `valid` 2x3 convs with stride (1,2) -> flatten -> dense layers; SELU everywhere.
// v1 layout: NHWC activations in HBM per batch of sites, one thread per output element; weights stay in
// the Keras layouts (HWIO / [in,out]) so consecutive lanes (output channels) read consecutive weights.
#include "nc_common.h"

namespace {

constexpr float SELU_L = 1.0507009873554805f;
constexpr float SELU_LA = 1.0507009873554805f * 1.6732632423543772f;

__device__ __forceinline__ float selu(float x) { return x > 0.0f ? SELU_L * x : SELU_LA * (expf(x) - 1.0f); }

// coverage scaling of rows 1.., channels 0..3 (snpCaller.py:93-96); see nanocaller_hip.h for the two modes
__device__ __forceinline__ float scaled_in(float x, const double *scale, int64_t s, int mode, int row, int ch, int Ci)
{
    if (!scale || row == 0 || ch == Ci - 1) return x;
    return mode == 0 ? x * (float)scale[s] : (float)((double)x * scale[s]);
}

struct ConvP {
    int H, W, Ci, kh, kw, Co, sh, sw, ph, pw, Ho, Wo, co_off, Ctot;
};

template <bool SCALE>
__global__ __launch_bounds__(256) void k_conv(const float *__restrict__ in, const float *__restrict__ k, const float *__restrict__ b,
                                              float *__restrict__ out, int64_t n_out_total, ConvP p,
                                              const double *__restrict__ scale, int scale_mode, int64_t site0)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_out_total) return;
    const int o = (int)(idx % p.Co);
    int64_t r = idx / p.Co;
    const int x = (int)(r % p.Wo);
    r /= p.Wo;
    const int y = (int)(r % p.Ho);
    const int64_t s = r / p.Ho;
    float acc = b[o];
    const float *ins = in + s * (int64_t)p.H * p.W * p.Ci;
    for (int dy = 0; dy < p.kh; dy++) {
        const int iy = y * p.sh + dy - p.ph;
        if (iy < 0 || iy >= p.H) continue;
        for (int dx = 0; dx < p.kw; dx++) {
            const int ix = x * p.sw + dx - p.pw;
            if (ix < 0 || ix >= p.W) continue;
            const float *ip = ins + ((int64_t)iy * p.W + ix) * p.Ci;
            const float *kp = k + ((int64_t)(dy * p.kw + dx) * p.Ci) * p.Co + o;
            for (int c = 0; c < p.Ci; c++) {
                float xv = ip[c];
                if (SCALE) xv = scaled_in(xv, scale, site0 + s, scale_mode, iy, c, p.Ci);
                acc = fmaf(xv, kp[(int64_t)c * p.Co], acc);
            }
        }
    }
    out[((s * p.Ho + y) * p.Wo + x) * p.Ctot + p.co_off + o] = selu(acc);
}

__global__ __launch_bounds__(256) void k_dense(const float *__restrict__ in, int n_in, const float *__restrict__ k,
                                               const float *__restrict__ b, int n_out, float *__restrict__ out, int64_t total,
                                               int act)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int o = (int)(idx % n_out);
    const int64_t s = idx / n_out;
    const float *ip = in + s * n_in;
    float acc = b[o];
    for (int i = 0; i < n_in; i++) acc = fmaf(ip[i], k[(int64_t)i * n_out + o], acc);
    out[idx] = act ? selu(acc) : acc;
}

__device__ __forceinline__ void dense_small(const float *in, int n_in, const float *k, const float *b, int n_out, float *out, bool act)
{
    for (int o = 0; o < n_out; o++) {
        float acc = b[o];
        for (int i = 0; i < n_in; i++) acc = fmaf(in[i], k[i * n_out + o], acc);
        out[o] = act ? selu(acc) : acc;
    }
}

__device__ __forceinline__ void softmax_small(float *x, int n)
{
    float m = x[0];
    for (int i = 1; i < n; i++) m = fmaxf(m, x[i]);
    float sum = 0.0f;
    for (int i = 0; i < n; i++) { x[i] = expf(x[i] - m); sum += x[i]; }
    for (int i = 0; i < n; i++) x[i] = x[i] / sum;
}

// SNP diploid tail: fa, four allele heads, fc2, fc3, GT (model_architect.py:53-62). w = pointer to fa.k
__global__ __launch_bounds__(256) void k_snp_heads(const float *__restrict__ fc1, const float *__restrict__ w,
                                                   const int32_t *__restrict__ ref_code, int64_t n, float *__restrict__ probs,
                                                   float *__restrict__ gt)
{
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    float f1[48], fa[16], in17[17], heads[4][2], fc2[16], in24[24], fc3[8], g[2];
    for (int i = 0; i < 48; i++) f1[i] = fc1[s * 48 + i];
    const float *p = w;
    dense_small(f1, 48, p, p + 768, 16, fa, true);
    p += 768 + 16;
    for (int i = 0; i < 16; i++) in17[i] = fa[i];
    const int rc = ref_code[s];
    for (int h = 0; h < 4; h++) {
        in17[16] = rc == h ? 1.0f : 0.0f;
        dense_small(in17, 17, p, p + 34, 2, heads[h], false);
        p += 34 + 2;
        softmax_small(heads[h], 2);
        probs[s * 4 + h] = heads[h][1];
    }
    dense_small(f1, 48, p, p + 768, 16, fc2, true);
    p += 768 + 16;
    for (int i = 0; i < 16; i++) in24[i] = fc2[i];
    for (int h = 0; h < 4; h++) { in24[16 + 2 * h] = heads[h][0]; in24[17 + 2 * h] = heads[h][1]; }
    dense_small(in24, 24, p, p + 192, 8, fc3, true);
    p += 192 + 8;
    dense_small(fc3, 8, p, p + 16, 2, g, false);
    softmax_small(g, 2);
    if (gt) { gt[s * 2] = g[0]; gt[s * 2 + 1] = g[1]; }
}

// haploid SNP tail: fc2, fc3 = Dense(4, selu) on [fc2, ref one-hot], softmax (model_architect_SNP_haploid.py:49-51)
__global__ __launch_bounds__(256) void k_snp_hap_heads(const float *__restrict__ fc1, const float *__restrict__ w,
                                                       const int32_t *__restrict__ ref_code, int64_t n, float *__restrict__ probs)
{
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    float f1[48], in20[20], out[4];
    for (int i = 0; i < 48; i++) f1[i] = fc1[s * 48 + i];
    const float *p = w;
    dense_small(f1, 48, p, p + 768, 16, in20, true);
    p += 768 + 16;
    const int rc = ref_code[s];
    for (int b = 0; b < 4; b++) in20[16 + b] = rc == b ? 1.0f : 0.0f;
    dense_small(in20, 20, p, p + 80, 4, out, true);
    softmax_small(out, 4);
    for (int b = 0; b < 4; b++) probs[s * 4 + b] = out[b];
}

// indel tail: fc2 (32->24 selu), fc3 (24->4 softmax | 24->1 sigmoid)
__global__ __launch_bounds__(256) void k_indel_heads(const float *__restrict__ fc1, const float *__restrict__ w, int nout, int64_t n,
                                                     float *__restrict__ probs)
{
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    float f1[32], f2[24], out[4];
    for (int i = 0; i < 32; i++) f1[i] = fc1[s * 32 + i];
    const float *p = w;
    dense_small(f1, 32, p, p + 768, 24, f2, true);
    p += 768 + 24;
    dense_small(f2, 24, p, p + 24 * nout, nout, out, false);
    if (nout == 4) {
        softmax_small(out, 4);
        for (int b = 0; b < 4; b++) probs[s * 4 + b] = out[b];
    } else {
        probs[s] = 1.0f / (1.0f + expf(-out[0]));
    }
}

struct Arch { int H, W, Ci, C1, C2, C3, F; };
const Arch ARCH[4] = {{5, 41, 5, 16, 32, 64, 48}, {5, 41, 5, 16, 32, 64, 48}, {15, 128, 2, 8, 32, 48, 32}, {5, 128, 2, 8, 32, 48, 32}};
const size_t NPARAM[4] = {109370, 108308, 634420, 158185};

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

// conv trunk for sites [site0, site0+nb) -> fc1 activations [nb][F] in ctx->cnn_c ; returns pointer to the tail weights
int run_trunk(nc_ctx *ctx, int kind, int64_t site0, int64_t nb, const float *x_batch, const double *scale, int scale_mode,
              const float **tail)
{
    const Arch A = ARCH[kind];
    const float *w = ctx->w[kind].dev;
    const int H2 = A.H - 1, W2 = (A.W - 3) / 2 + 1, H3 = H2 - 1, W3 = (W2 - 3) / 2 + 1;
    const int64_t n1 = (int64_t)A.H * A.W * 3 * A.C1, n2 = (int64_t)H2 * W2 * A.C2, n3 = (int64_t)H3 * W3 * A.C3;
    NC_TRY(nc_ensure(ctx, ctx->cnn_a, (size_t)(nb * n1) * 4));
    NC_TRY(nc_ensure(ctx, ctx->cnn_b, (size_t)(nb * n2) * 4));
    NC_TRY(nc_ensure(ctx, ctx->cnn_c, (size_t)(nb * (n3 + A.F)) * 4));
    float *a1 = (float *)ctx->cnn_a.p, *a2 = (float *)ctx->cnn_b.p, *a3 = (float *)ctx->cnn_c.p, *f1 = a3 + nb * n3;
    const float *k11 = w, *b11 = k11 + 1 * 5 * A.Ci * A.C1;
    const float *k12 = b11 + A.C1, *b12 = k12 + 5 * 1 * A.Ci * A.C1;
    const float *k13 = b12 + A.C1, *b13 = k13 + 5 * 5 * A.Ci * A.C1;
    const float *k2 = b13 + A.C1, *b2 = k2 + 2 * 3 * 3 * A.C1 * A.C2;
    const float *k3 = b2 + A.C2, *b3 = k3 + 2 * 3 * A.C2 * A.C3;
    const float *kf = b3 + A.C3, *bf = kf + n3 * A.F;
    *tail = bf + A.F;
    ConvP p;
    p.H = A.H; p.W = A.W; p.Ci = A.Ci; p.Co = A.C1; p.sh = 1; p.sw = 1; p.Ho = A.H; p.Wo = A.W; p.Ctot = 3 * A.C1;
    const int64_t tot1 = nb * A.H * A.W * A.C1;
    p.kh = 1; p.kw = 5; p.ph = 0; p.pw = 2; p.co_off = 0;
    hipLaunchKernelGGL(k_conv<true>, dim3(blocks_for(tot1)), dim3(256), 0, ctx->stream, x_batch, k11, b11, a1, tot1, p, scale, scale_mode, site0);
    p.kh = 5; p.kw = 1; p.ph = 2; p.pw = 0; p.co_off = A.C1;
    hipLaunchKernelGGL(k_conv<true>, dim3(blocks_for(tot1)), dim3(256), 0, ctx->stream, x_batch, k12, b12, a1, tot1, p, scale, scale_mode, site0);
    p.kh = 5; p.kw = 5; p.ph = 2; p.pw = 2; p.co_off = 2 * A.C1;
    hipLaunchKernelGGL(k_conv<true>, dim3(blocks_for(tot1)), dim3(256), 0, ctx->stream, x_batch, k13, b13, a1, tot1, p, scale, scale_mode, site0);
    ConvP q;
    q.H = A.H; q.W = A.W; q.Ci = 3 * A.C1; q.kh = 2; q.kw = 3; q.Co = A.C2; q.sh = 1; q.sw = 2; q.ph = 0; q.pw = 0;
    q.Ho = H2; q.Wo = W2; q.co_off = 0; q.Ctot = A.C2;
    hipLaunchKernelGGL(k_conv<false>, dim3(blocks_for(nb * n2)), dim3(256), 0, ctx->stream, a1, k2, b2, a2, nb * n2, q, nullptr, 0, (int64_t)0);
    ConvP r;
    r.H = H2; r.W = W2; r.Ci = A.C2; r.kh = 2; r.kw = 3; r.Co = A.C3; r.sh = 1; r.sw = 2; r.ph = 0; r.pw = 0;
    r.Ho = H3; r.Wo = W3; r.co_off = 0; r.Ctot = A.C3;
    hipLaunchKernelGGL(k_conv<false>, dim3(blocks_for(nb * n3)), dim3(256), 0, ctx->stream, a2, k3, b3, a3, nb * n3, r, nullptr, 0, (int64_t)0);
    hipLaunchKernelGGL(k_dense, dim3(blocks_for(nb * A.F)), dim3(256), 0, ctx->stream, a3, (int)n3, kf, bf, A.F, f1, nb * A.F, 1);
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

}   // namespace

extern "C" {

int nc_load_weights(nc_ctx *ctx, int32_t kind, const float *blob_host, size_t n_floats)
{
    if (!ctx) return NC_ERR_ARG;
    if (kind < 0 || kind > 3 || !blob_host) return nc_fail(ctx, NC_ERR_ARG, "nc_load_weights: bad argument");
    if (n_floats != NPARAM[kind])
        return nc_fail(ctx, NC_ERR_ARG, "nc_load_weights: kind %d expects %zu floats, got %zu", kind, NPARAM[kind], n_floats);
    NC_HIP(ctx, hipSetDevice(ctx->device));
    nc_weights &w = ctx->w[kind];
    if (!w.dev) {
        hipError_t e = hipMalloc(&w.dev, n_floats * 4);
        if (e != hipSuccess) return nc_fail(ctx, NC_ERR_NOMEM, "hipMalloc weights: %s", hipGetErrorString(e));
    }
    NC_HIP(ctx, hipMemcpyAsync(w.dev, blob_host, n_floats * 4, hipMemcpyHostToDevice, ctx->stream));
    NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    w.n = n_floats;
    return NC_OK;
}

int nc_snp_forward(nc_ctx *ctx, int32_t kind, int64_t n, const float *x_dev, const int32_t *ref_code_dev, const double *scale_dev,
                   int32_t scale_mode, float *probs_dev, float *gt_dev)
{
    if (!ctx) return NC_ERR_ARG;
    if (kind != NC_MODEL_SNP && kind != NC_MODEL_SNP_HAP) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_forward: not an SNP model kind");
    if (!ctx->w[kind].dev) return nc_fail(ctx, NC_ERR_STATE, "nc_snp_forward: weights of kind %d not loaded", kind);
    if (n < 0 || (n && (!x_dev || !ref_code_dev || !probs_dev))) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_forward: null argument");
    if (scale_mode != 0 && scale_mode != 1) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_forward: scale_mode");
    NC_HIP(ctx, hipSetDevice(ctx->device));
    NcTimer tm(ctx, 2);
    const int64_t BATCH = 16384;
    for (int64_t s0 = 0; s0 < n; s0 += BATCH) {
        const int64_t nb = n - s0 < BATCH ? n - s0 : BATCH;
        const float *tail = nullptr;
        NC_TRY(run_trunk(ctx, kind, s0, nb, x_dev + s0 * NC_SNP_TENSOR, scale_dev, scale_mode, &tail));
        const float *f1 = (const float *)ctx->cnn_c.p + nb * (3 * 9 * 64);
        if (kind == NC_MODEL_SNP)
            hipLaunchKernelGGL(k_snp_heads, dim3(blocks_for(nb)), dim3(256), 0, ctx->stream, f1, tail, ref_code_dev + s0, nb,
                               probs_dev + s0 * 4, gt_dev ? gt_dev + s0 * 2 : nullptr);
        else
            hipLaunchKernelGGL(k_snp_hap_heads, dim3(blocks_for(nb)), dim3(256), 0, ctx->stream, f1, tail, ref_code_dev + s0, nb,
                               probs_dev + s0 * 4);
        NC_HIP(ctx, hipGetLastError());
    }
    tm.stop();
    return NC_OK;
}

int nc_indel_forward(nc_ctx *ctx, int32_t kind, int64_t n, const float *x_dev, float *probs_dev)
{
    if (!ctx) return NC_ERR_ARG;
    if (kind != NC_MODEL_INDEL && kind != NC_MODEL_INDEL_HAP) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_forward: not an indel model kind");
    if (!ctx->w[kind].dev) return nc_fail(ctx, NC_ERR_STATE, "nc_indel_forward: weights of kind %d not loaded", kind);
    if (n < 0 || (n && (!x_dev || !probs_dev))) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_forward: null argument");
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const Arch A = ARCH[kind];
    const int nout = kind == NC_MODEL_INDEL ? 4 : 1;
    const int64_t xs = (int64_t)A.H * A.W * A.Ci;
    const int H3 = A.H - 2, W3 = 31;
    NcTimer tm(ctx, 2);
    const int64_t BATCH = kind == NC_MODEL_INDEL ? 2048 : 8192;
    for (int64_t s0 = 0; s0 < n; s0 += BATCH) {
        const int64_t nb = n - s0 < BATCH ? n - s0 : BATCH;
        const float *tail = nullptr;
        NC_TRY(run_trunk(ctx, kind, s0, nb, x_dev + s0 * xs, nullptr, 0, &tail));
        const float *f1 = (const float *)ctx->cnn_c.p + nb * ((int64_t)H3 * W3 * A.C3);
        hipLaunchKernelGGL(k_indel_heads, dim3(blocks_for(nb)), dim3(256), 0, ctx->stream, f1, tail, nout, nb, probs_dev + s0 * nout);
        NC_HIP(ctx, hipGetLastError());
    }
    tm.stop();
    return NC_OK;
}

}   // extern "C"
