// K5 / K9: CNN forward for the four NanoCaller models (gfx950), exact fp32.
//
// Restates model_architect.py:36-64, model_architect_SNP_haploid.py:33-53, model_architect_indel.py:28-48,
// model_architect_indels_haploid.py:29-48: three parallel `same` convs (1x5, 5x1, 5x5)
// -> concat -> two `valid` 2x3 convs with stride (1,2) -> flatten -> dense layers; SELU everywhere.
//
// Kernel shape ("scalar-weight direct convolution"): one lane owns one output POSITION (site, y, x) and keeps
// ALL output channels of that position in VGPR accumulators; the K loop walks (tap, input channel); the
// activation is a per-lane value (dwordx4 loads of the NHWC row), the weight row w[tap][ci][0..Co) is
// wave-uniform and arrives through the scalar cache, so every v_fma_f32 takes one VGPR activation and one SGPR
// weight: no LDS staging, no im2col, and the arithmetic is an fmaf chain in the reference's (tap, ci) order.
// fp32 matrix and vector peaks are equal on gfx950 (157.3 TFLOP/s), so this VALU form has the same roof as
// v_mfma_f32_* while keeping the weights out of the vector register file.
#include <cmath>
#include <cstdlib>
#include <vector>

#include <hip/hip_ext.h>

#include "nc_common.h"

namespace {

constexpr float SELU_L = 1.0507009873554805f;
constexpr float SELU_LA = 1.0507009873554805f * 1.6732632423543772f;

// SELU.  The trunk's 14k activations per site use the hardware exponential (v_exp_f32 after a multiply by log2 e,
// relative error ~1e-6 at most for x in [-20, 0]: absolute error of the negative branch < 2e-6); the tiny heads
// use the accurate expf.  Parity tests hold the end-to-end probabilities far inside 1e-4.
__device__ __forceinline__ float selu(float x) { return x > 0.0f ? SELU_L * x : SELU_LA * (__expf(x) - 1.0f); }
// branchless form for the split-precision trunk (one v_exp + select; no exec-mask branch in the epilogues)
__device__ __forceinline__ float selu_bl(float x)
{
    const float e = SELU_LA * (__expf(fminf(x, 0.0f)) - 1.0f);
    return x > 0.0f ? SELU_L * x : e;
}
__device__ __forceinline__ float selu_acc(float x) { return x > 0.0f ? SELU_L * x : SELU_LA * (expf(x) - 1.0f); }

// ---- conv1: the three `same` convolutions, fused.  Canonical weights: k11[1][5][CI][C1] b11 k12[5][1][CI][C1] b12
// k13[5][5][CI][C1] b13.  Output NHWC [site][H][W][3*C1].  Coverage scaling (snpCaller.py:93-96) is applied while
// loading: rows >= 1, channels < CI-1 (scale == nullptr: none).
template <int H, int W, int CI, int C1>
__global__ __launch_bounds__(256) void k2_conv1(const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ out,
                                                int64_t npos, const double *__restrict__ scale, int scale_mode, int64_t site0)
{
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= npos) return;
    const int64_t site = g / (H * W);
    const int r = (int)(g - site * (H * W));
    const int h = r / W, wq = r - h * W;
    const float *xs = x + site * (H * W * CI);
    const float *k11 = w, *b11 = k11 + 5 * CI * C1;
    const float *k12 = b11 + C1, *b12 = k12 + 5 * CI * C1;
    const float *k13 = b12 + C1, *b13 = k13 + 25 * CI * C1;
    float a1[C1], a2[C1], a3[C1];
#pragma unroll
    for (int o = 0; o < C1; o++) { a1[o] = b11[o]; a2[o] = b12[o]; a3[o] = b13[o]; }
    float sf = 1.0f;
    double sd = 1.0;
    if (scale) { sd = scale[site0 + site]; sf = (float)sd; }
#pragma unroll 1
    for (int dy = -2; dy <= 2; dy++) {
#pragma unroll 1
        for (int dx = -2; dx <= 2; dx++) {
            const int iy = h + dy, ix = wq + dx;
            const bool inb = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float *ip = xs + (iy * W + ix) * CI;
#pragma unroll
            for (int c = 0; c < CI; c++) {
                float xv = inb ? ip[c] : 0.0f;
                if (scale && c < CI - 1 && iy > 0) xv = scale_mode == 0 ? xv * sf : (float)((double)xv * sd);
                const float *w3 = k13 + (((dy + 2) * 5 + (dx + 2)) * CI + c) * C1;
#pragma unroll
                for (int o = 0; o < C1; o++) a3[o] = fmaf(xv, w3[o], a3[o]);
                if (dy == 0) {
                    const float *w1 = k11 + ((dx + 2) * CI + c) * C1;
#pragma unroll
                    for (int o = 0; o < C1; o++) a1[o] = fmaf(xv, w1[o], a1[o]);
                }
                if (dx == 0) {
                    const float *w2 = k12 + ((dy + 2) * CI + c) * C1;
#pragma unroll
                    for (int o = 0; o < C1; o++) a2[o] = fmaf(xv, w2[o], a2[o]);
                }
            }
        }
    }
    float4 *op = reinterpret_cast<float4 *>(out + g * (3 * C1));
#pragma unroll
    for (int o = 0; o < C1; o += 4) {
        op[o / 4] = make_float4(selu(a1[o]), selu(a1[o + 1]), selu(a1[o + 2]), selu(a1[o + 3]));
        op[(C1 + o) / 4] = make_float4(selu(a2[o]), selu(a2[o + 1]), selu(a2[o + 2]), selu(a2[o + 3]));
        op[(2 * C1 + o) / 4] = make_float4(selu(a3[o]), selu(a3[o + 1]), selu(a3[o + 2]), selu(a3[o + 3]));
    }
}

// conv1 of the indel models (CI = 2, W a multiple of 4), four x-adjacent output positions per thread: the 5 x 8 input
// window of the four positions is loaded once (20 dwordx4 instead of 100 8-byte loads) and every weight, a wave-uniform
// scalar operand, feeds four FMAs.  Same fmaf order per output as k2_conv1 (tap-major, channel-minor): bit-identical.
// SPLIT: the activations leave as two fp16 planes (hi = fp16(v), lo = fp16(v - hi); same bytes as fp32), the operand form
// of the split-precision conv2 (k8_conv23_h3): `out` = hi plane [npos][3*C1], the lo plane follows it.
template <int H, int W, int C1, bool SPLIT>
__global__ __launch_bounds__(256) void k2_conv1_x4(const float *__restrict__ x, const float *__restrict__ w, float *__restrict__ out, int64_t npos)
{
    static_assert(W % 4 == 0 && C1 == 8, "k2_conv1_x4: shape");
    constexpr int CI = 2;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t * 4 >= npos) return;
    const int64_t g0 = t * 4, site = g0 / (H * W);
    const int r = (int)(g0 - site * (H * W));
    const int h = r / W, x0 = r - h * W;                                  // x0 % 4 == 0
    const float *xs = x + site * (H * W * CI);
    const float *k11 = w, *b11 = k11 + 5 * CI * C1;
    const float *k12 = b11 + C1, *b12 = k12 + 5 * CI * C1;
    const float *k13 = b12 + C1, *b13 = k13 + 25 * CI * C1;
    float a1[4][C1], a2[4][C1], a3[4][C1];
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
        for (int o = 0; o < C1; o++) { a1[p][o] = b11[o]; a2[p][o] = b12[o]; a3[p][o] = b13[o]; }
#pragma unroll 1
    for (int dy = -2; dy <= 2; dy++) {
        const int iy = h + dy;
        // pixels x0-2 .. x0+5 of row iy, two channels each: 16 floats (zero outside the image)
        float win[16];
        const bool row_in = iy >= 0 && iy < H;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int px = x0 - 2 + 2 * q;                                 // pixel pair (px, px + 1): both inside or both outside
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (row_in && px >= 0 && px + 1 < W) v = *reinterpret_cast<const float4 *>(xs + ((int64_t)iy * W + px) * CI);
            win[4 * q] = v.x; win[4 * q + 1] = v.y; win[4 * q + 2] = v.z; win[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int dx = -2; dx <= 2; dx++) {
#pragma unroll
            for (int c = 0; c < CI; c++) {
                const float *w3 = k13 + (((dy + 2) * 5 + (dx + 2)) * CI + c) * C1;
#pragma unroll
                for (int p = 0; p < 4; p++) {
                    const float xv = win[(p + dx + 2) * CI + c];
#pragma unroll
                    for (int o = 0; o < C1; o++) a3[p][o] = fmaf(xv, w3[o], a3[p][o]);
                }
                if (dy == 0) {
                    const float *w1 = k11 + ((dx + 2) * CI + c) * C1;
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const float xv = win[(p + dx + 2) * CI + c];
#pragma unroll
                        for (int o = 0; o < C1; o++) a1[p][o] = fmaf(xv, w1[o], a1[p][o]);
                    }
                }
                if (dx == 0) {
                    const float *w2 = k12 + ((dy + 2) * CI + c) * C1;
#pragma unroll
                    for (int p = 0; p < 4; p++) {
                        const float xv = win[(p + 2) * CI + c];
#pragma unroll
                        for (int o = 0; o < C1; o++) a2[p][o] = fmaf(xv, w2[o], a2[p][o]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const int64_t g = g0 + p;
        if constexpr (SPLIT) {
            _Float16 *hp = reinterpret_cast<_Float16 *>(out) + g * (3 * C1), *lp = hp + npos * (3 * C1);
            const float *acc[3] = {a1[p], a2[p], a3[p]};
#pragma unroll
            for (int b = 0; b < 3; b++) {
                _Float16 hi[8], lo[8];
#pragma unroll
                for (int o = 0; o < 8; o++) {
                    const float v = fminf(fmaxf(selu(acc[b][o]), -65504.0f), 65504.0f);
                    hi[o] = (_Float16)v;
                    lo[o] = (_Float16)(v - (float)hi[o]);
                }
                *reinterpret_cast<uint4 *>(hp + 8 * b) = *reinterpret_cast<const uint4 *>(hi);
                *reinterpret_cast<uint4 *>(lp + 8 * b) = *reinterpret_cast<const uint4 *>(lo);
            }
        } else {
            float4 *op = reinterpret_cast<float4 *>(out + g * (3 * C1));
#pragma unroll
            for (int o = 0; o < C1; o += 4) {
                op[o / 4] = make_float4(selu(a1[p][o]), selu(a1[p][o + 1]), selu(a1[p][o + 2]), selu(a1[p][o + 3]));
                op[(C1 + o) / 4] = make_float4(selu(a2[p][o]), selu(a2[p][o + 1]), selu(a2[p][o + 2]), selu(a2[p][o + 3]));
                op[(2 * C1 + o) / 4] = make_float4(selu(a3[p][o]), selu(a3[p][o + 1]), selu(a3[p][o + 2]), selu(a3[p][o + 3]));
            }
        }
    }
}

// ---- MFMA forms (SNP trunk).  fp32-in/fp32-accumulate MFMA is bit-for-bit an fmaf chain (exact fp32).
// GEMM view: M = output positions or sites (A fragment: one activation per lane), N = output channels (B fragment:
// one weight per lane), K = (tap, ci) walked in a permuted order so that the 4 consecutive input channels a lane
// loads as ONE dwordx4 feed 4 consecutive MFMAs.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// fc1 as v_mfma_f32_16x16x4: M = 16 sites per tile, N = F/16 tiles, K walked in groups of 16 (quarter-wave q takes
// k in [16j+4q, 16j+4q+4) as one dwordx4).  lane l: A[row l&15][k l>>4], B[k l>>4][col l&15]; C: col l&15, row 4*(l>>4)+r.
// Split-K: the four waves of a workgroup share the same 16*TM sites and each takes a quarter of K; partial sums are
// combined through LDS (K = 1728 would otherwise be one 40k-cycle dependent chain per wave).
template <int F, int TM>
__global__ __launch_bounds__(256) void k3_fc1(const float *__restrict__ in, int K, const float *__restrict__ wk, const float *__restrict__ wb,
                                              float *__restrict__ out, int64_t n)
{
    constexpr int TN = F / 16;
    __shared__ float red[3][TM][TN][4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = lane >> 4, c16 = lane & 15;
    const int64_t tile0 = (int64_t)blockIdx.x * (TM * 16);
    const float *ip[TM];
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
        int64_t s = tile0 + tm * 16 + c16;
        if (s >= n) s = n - 1;
        ip[tm] = in + s * K + 4 * q;
    }
    f32x4v acc[TM][TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
        const float b = wv == 0 ? wb[tn * 16 + c16] : 0.0f;
#pragma unroll
        for (int tm = 0; tm < TM; tm++) acc[tm][tn] = (f32x4v){b, b, b, b};
    }
    const float *wl = wk + (4 * q) * F + c16;
    const int ng = K / 16;
    const int j0 = (ng * wv) / 4, j1 = (ng * (wv + 1)) / 4;
    // The loads of group j + NS - 1 are issued before the MFMAs of group j: a wave keeps NS - 1 groups (activations from HBM,
    // weights from L2) in flight instead of waiting for each group's loads with nothing behind them.
#ifdef NC_K3_NS
    constexpr int NS = NC_K3_NS;
#else
    constexpr int NS = TM <= 2 ? 4 : 2;
#endif
    float4 a[NS][TM];
    float b[NS][4][TN];
    auto ld = [&](int st, int j) {
        j = min(j, j1 - 1);                                             // (past the end: the last group again, unused)
#pragma unroll
        for (int tm = 0; tm < TM; tm++) a[st][tm] = *reinterpret_cast<const float4 *>(ip[tm] + 16 * j);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int tn = 0; tn < TN; tn++) b[st][i][tn] = wl[(int64_t)(16 * j + i) * F + tn * 16];
    };
    if (j0 < j1) {
#pragma unroll
        for (int st = 0; st < NS - 1; st++) ld(st, j0 + st);
    }
    auto mm = [&](int u) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int tm = 0; tm < TM; tm++) {
                const float av = i == 0 ? a[u][tm].x : i == 1 ? a[u][tm].y : i == 2 ? a[u][tm].z : a[u][tm].w;
#pragma unroll
                for (int tn = 0; tn < TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b[u][i][tn], acc[tm][tn], 0, 0, 0);
            }
    };
    int j = j0;
    for (; j + NS <= j1; j += NS) {                                     // whole rounds: no branch between a load and its use
#pragma unroll
        for (int u = 0; u < NS; u++) {
            ld((u + NS - 1) % NS, j + u + NS - 1);
            __builtin_amdgcn_sched_barrier(0);                          // (the scheduler otherwise sinks these loads below the products)
            mm(u);
        }
    }
#pragma unroll
    for (int u = 0; u < NS - 1; u++)                                    // the last groups are already on their way (stage u = group j + u)
        if (j + u < j1) mm(u);
    if (wv > 0) {
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
            for (int tn = 0; tn < TN; tn++)
#pragma unroll
                for (int r = 0; r < 4; r++) red[wv - 1][tm][tn][r][lane] = acc[tm][tn][r];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t s = tile0 + tm * 16 + 4 * q + r;
#pragma unroll
                for (int tn = 0; tn < TN; tn++) {
                    const float v = acc[tm][tn][r] + red[0][tm][tn][r][lane] + red[1][tm][tn][r][lane] + red[2][tm][tn][r][lane];
                    if (s < n) out[s * F + tn * 16 + c16] = selu(v);
                }
            }
        }
    }
}


// ---- conv2 / conv3 of the indel models as an implicit GEMM on exact-fp32 MFMA (v_mfma_f32_16x16x4_f32, bit-for-bit an
// fmaf chain): M = output positions (one tile of 16 per wave iteration), N = CO, K = 6*CI (tap-major) walked in groups of
// 16.  A lane's float4 = 4 consecutive input channels of one tap and feeds 4 MFMA steps: K slot kq of step j is
// k = 16 G + 4 kq + j, for the activation and the weight operand alike.  The weights of the layer live in LDS in fragment
// order (one ds_read_b128 per 4 MFMAs).  in NHWC [site][HI][WI][CI], weights [2][3][CI][CO], out NHWC [site][HO][WO][CO].
template <int HI, int WI, int CI, int CO>
__global__ __launch_bounds__(256) void k7_conv23_mfma(const float *__restrict__ in, const float *__restrict__ wk, const float *__restrict__ wb,
                                                      float *__restrict__ out, int64_t npos)
{
    constexpr int HO = HI - 1, WO = (WI - 3) / 2 + 1, K = 6 * CI, NG = K / 16, TN = CO / 16;
    static_assert(CI % 4 == 0 && K % 16 == 0 && CO % 16 == 0, "k7_conv23_mfma: shape");
    __shared__ float4 wf[NG][TN][64];
    for (int idx = threadIdx.x; idx < NG * TN * 64; idx += 256) {
        const int l = idx & 63, tn = (idx >> 6) % TN, G = (idx >> 6) / TN;
        const int k0 = 16 * G + 4 * (l >> 4), col = tn * 16 + (l & 15);
        wf[G][tn][l] = make_float4(wk[(k0 + 0) * CO + col], wk[(k0 + 1) * CO + col], wk[(k0 + 2) * CO + col], wk[(k0 + 3) * CO + col]);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, kq = lane >> 4, c16 = lane & 15;
    const int64_t ntiles = (npos + 15) / 16;
    float bias[TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) bias[tn] = wb[tn * 16 + c16];
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        int64_t m = tile * 16 + c16;
        if (m >= npos) m = npos - 1;
        const int64_t site = m / (HO * WO);
        const int r = (int)(m - site * (HO * WO));
        const int y = r / WO, xq = r - y * WO;
        const float *ip = in + ((site * HI + y) * WI + 2 * xq) * CI;
        f32x4v acc[TN];
#pragma unroll
        for (int tn = 0; tn < TN; tn++) acc[tn] = (f32x4v){bias[tn], bias[tn], bias[tn], bias[tn]};
#pragma unroll
        for (int G = 0; G < NG; G++) {
            const int k0 = 16 * G + 4 * kq, tap = k0 / CI, ci = k0 - tap * CI;
            const float4 a = *reinterpret_cast<const float4 *>(ip + ((tap / 3) * WI + (tap % 3)) * CI + ci);
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                const float4 b = wf[G][tn][lane];
                acc[tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[tn], 0, 0, 0);
                acc[tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[tn], 0, 0, 0);
                acc[tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[tn], 0, 0, 0);
                acc[tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[tn], 0, 0, 0);
            }
        }
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int64_t pos = tile * 16 + 4 * kq + rr;             // D[row 4 kq + rr = position][col c16 = channel]
            if (pos < npos) {
#pragma unroll
                for (int tn = 0; tn < TN; tn++) out[pos * CO + tn * 16 + c16] = selu(acc[tn][rr]);
            }
        }
    }
}

// ---- fused conv1 + conv2 for the SNP trunk (5x41x5 input): the 205x48 conv1 activation lives only in LDS.
// Per site: (1) the input is staged, coverage-scaled, into a zero-padded [9][45][5] LDS image; (2) conv1 runs as
// v_mfma_f32_16x16x4 over 13 tiles of 16 positions: K is laid out as 5 input rows x 28 (25 real (dx,ci) values, which
// are CONTIGUOUS in the NHWC image, + 3 zero-weight slots) = 35 steps for the 5x5 kernel; the 1x5 kernel reuses the
// A fragments of row dy=2 (7 more MFMAs), the 5x1 kernel those of steps ls=2,3 of every row (10 more MFMAs, zero
// weights outside dx=2): 52 MFMAs per tile, 35 ds_read_b32; (3) conv2 (2x3, stride (1,2)) reads its A fragments from
// the LDS activation with one ds_read_b128 per 4 MFMAs.  Weight (B) fragments are pre-packed in fragment order
// (one coalesced 256-B read per MFMA).  lane l: A[row l&15][k l>>4], B[k l>>4][col l&15], C/D col l&15, row 4*(l>>4)+r.
constexpr int F12_XP = 2032;             // padded input image (9*45*5 = 2025, +7 so zero-weight slots stay in range)
constexpr int F12_CP = 52;               // channel pitch of the LDS activation (48 + 4: keeps ds_read_b128 aligned, spreads banks)
constexpr int F12_W1P = 52 * 64;         // conv1 B fragments
constexpr int F12_W2P = 6 * 3 * 4 * 2 * 64;
constexpr int F12_W3P = 48 * 4 * 64;      // conv3 B fragments: [step 48][tn 4][lane 64]
constexpr int F12_CP2 = 36;               // channel pitch of the LDS conv2 activation (32 + 4)
constexpr int F12_PACKED = F12_W1P + 48 + F12_W2P + 32 + F12_W3P + 64;

template <int NT>
__device__ __forceinline__ void f12_conv1_pass(const float *Xp, float *A1, const float (&w1r)[52], const float *__restrict__ b1,
                                               int tile_first, int lane)
{
    const int kq = lane >> 4, c16 = lane & 15;
    int rowbase[NT];
#pragma unroll
    for (int tm = 0; tm < NT; tm++) {
        int p = (tile_first + 4 * tm) * 16 + c16;
        p = p < 205 ? p : 204;
        const int h = p / 41, w = p - h * 41;
        rowbase[tm] = (h * 45 + w) * 5 + kq;
    }
    f32x4v acc1[NT], acc2[NT], acc3[NT];
    {
        const float x1 = b1[c16], x2 = b1[16 + c16], x3 = b1[32 + c16];
#pragma unroll
        for (int tm = 0; tm < NT; tm++) {
            acc1[tm] = (f32x4v){x1, x1, x1, x1};
            acc2[tm] = (f32x4v){x2, x2, x2, x2};
            acc3[tm] = (f32x4v){x3, x3, x3, x3};
        }
    }
#pragma unroll
    for (int s = 0; s < 35; s++) {
        const int dy = s / 7, ls = s % 7;
        float a[NT];
#pragma unroll
        for (int tm = 0; tm < NT; tm++) a[tm] = Xp[rowbase[tm] + dy * 225 + 4 * ls];
#pragma unroll
        for (int tm = 0; tm < NT; tm++) acc3[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm], w1r[s], acc3[tm], 0, 0, 0);
        if (dy == 2) {
#pragma unroll
            for (int tm = 0; tm < NT; tm++) acc1[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm], w1r[35 + ls], acc1[tm], 0, 0, 0);
        }
        if (ls == 2 || ls == 3) {
#pragma unroll
            for (int tm = 0; tm < NT; tm++)
                acc2[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tm], w1r[42 + dy * 2 + (ls - 2)], acc2[tm], 0, 0, 0);
        }
    }
#pragma unroll
    for (int tm = 0; tm < NT; tm++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int pos = (tile_first + 4 * tm) * 16 + 4 * kq + r;
            if (pos < 205) {
                float *o = A1 + pos * F12_CP + c16;
                o[0] = selu(acc1[tm][r]);
                o[16] = selu(acc2[tm][r]);
                o[32] = selu(acc3[tm][r]);
            }
        }
    }
}

template <int NT>
__device__ __forceinline__ void f12_conv2(const float *A1, const float (&w2r)[72], const float *__restrict__ b2,
                                          float *A2, int wv, int lane)
{
    const int kq = lane >> 4, c16 = lane & 15, tn = wv & 1, t0 = wv >> 1;
    int abase[NT];
#pragma unroll
    for (int tm = 0; tm < NT; tm++) {
        const int p = (t0 + 2 * tm) * 16 + c16;          // < 80
        const int y = p / 20, x = p - y * 20;
        abase[tm] = (y * 41 + 2 * x) * F12_CP + 4 * kq;
    }
    f32x4v acc[NT];
    {
        const float b = b2[tn * 16 + c16];
#pragma unroll
        for (int tm = 0; tm < NT; tm++) acc[tm] = (f32x4v){b, b, b, b};
    }
#pragma unroll
    for (int tap = 0; tap < 6; tap++) {
        const int toff = ((tap / 3) * 41 + (tap % 3)) * F12_CP;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float4 a[NT];
#pragma unroll
            for (int tm = 0; tm < NT; tm++) a[tm] = *reinterpret_cast<const float4 *>(A1 + abase[tm] + toff + 16 * j);
#pragma unroll
            for (int i = 0; i < 4; i++) {
#pragma unroll
                for (int tm = 0; tm < NT; tm++) {
                    const float av = i == 0 ? a[tm].x : i == 1 ? a[tm].y : i == 2 ? a[tm].z : a[tm].w;
                    acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, w2r[(tap * 3 + j) * 4 + i], acc[tm], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int tm = 0; tm < NT; tm++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int po = (t0 + 2 * tm) * 16 + 4 * kq + r;
            A2[po * F12_CP2 + tn * 16 + c16] = selu(acc[tm][r]);
        }
    }
}

// conv3 (2x3, stride (1,2): 4x20x32 -> 3x9x64) out of the LDS conv2 activation: 27 positions = 2 tiles of 16, wave w
// owns output channels [16w, 16w+16); its 48 weight fragments are streamed from L2 (the register file is full).
__device__ __forceinline__ void f12_conv3(const float *A2, const float *__restrict__ w3p, const float *__restrict__ b3,
                                          float *__restrict__ out_site, int wv, int lane)
{
    const int kq = lane >> 4, c16 = lane & 15;
    int abase[2];
#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
        int p = tm * 16 + c16;
        p = p < 27 ? p : 26;
        const int y = p / 9, x = p - y * 9;
        abase[tm] = (y * 20 + 2 * x) * F12_CP2 + 4 * kq;
    }
    f32x4v acc[2];
    {
        const float b = b3[wv * 16 + c16];
        acc[0] = (f32x4v){b, b, b, b};
        acc[1] = acc[0];
    }
    const float *wl = w3p + wv * 64 + lane;
#pragma unroll 1
    for (int tap = 0; tap < 6; tap++) {
        const int toff = ((tap / 3) * 20 + (tap % 3)) * F12_CP2;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            float4 a[2];
#pragma unroll
            for (int tm = 0; tm < 2; tm++) a[tm] = *reinterpret_cast<const float4 *>(A2 + abase[tm] + toff + 16 * j);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float b = wl[((tap * 2 + j) * 4 + i) * 256];
#pragma unroll
                for (int tm = 0; tm < 2; tm++) {
                    const float av = i == 0 ? a[tm].x : i == 1 ? a[tm].y : i == 2 ? a[tm].z : a[tm].w;
                    acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[tm], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int po = tm * 16 + 4 * kq + r;
            if (po < 27) out_site[po * 64 + wv * 16 + c16] = selu(acc[tm][r]);
        }
    }
}

// Weight-stationary and persistent: every wave loads its 52 conv1 and 72 conv2 weight fragments into registers ONCE
// (124 VGPRs) and then walks sites; in steady state the only memory traffic is the 4.1 KB input tensor in, the 10 KB
// conv2 activation out, and LDS.  Two workgroups (8 waves) per CU.
__global__ __launch_bounds__(256, 2) void k4_conv12(const float *__restrict__ x, const float *__restrict__ wp, float *__restrict__ a3,
                                                    int64_t n_sites, const double *__restrict__ scale, int scale_mode, int64_t site0)
{
    __shared__ __attribute__((aligned(16))) float Xp[F12_XP];
    __shared__ __attribute__((aligned(16))) float A1[205 * F12_CP];
    __shared__ __attribute__((aligned(16))) float A2[80 * F12_CP2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float *w1p = wp, *b1 = wp + F12_W1P, *w2p = b1 + 48, *b2 = w2p + F12_W2P, *w3p = b2 + 32, *b3 = w3p + F12_W3P;
    float w1r[52], w2r[72];
#pragma unroll
    for (int s = 0; s < 52; s++) w1r[s] = w1p[s * 64 + lane];
#pragma unroll
    for (int s = 0; s < 72; s++) w2r[s] = w2p[(s * 2 + (wv & 1)) * 64 + lane];
    for (int i = threadIdx.x; i < F12_XP; i += 256) Xp[i] = 0.0f;
    __syncthreads();
    // staging is split: the global loads of the NEXT site are issued at the top of an iteration (5 values per thread,
    // held in registers while conv1 runs) and written, scaled, into the padded LDS image once conv1 has released it
    float pre[5];
    float pre_sf = 1.0f;
    double pre_sd = 1.0;
    auto prefetch = [&](int64_t site) {
        const float *xs = x + site * NC_SNP_TENSOR;
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int i = threadIdx.x + u * 256;
            pre[u] = i < NC_SNP_TENSOR ? xs[i] : 0.0f;
        }
        if (scale) { pre_sd = scale[site0 + site]; pre_sf = (float)pre_sd; }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 5; u++) {
            const int i = threadIdx.x + u * 256;
            if (i < NC_SNP_TENSOR) {
                const int h = i / 205, rem = i - h * 205, w = rem / 5, c = rem - w * 5;
                float v = pre[u];
                if (scale && h > 0 && c < 4) v = scale_mode == 0 ? v * pre_sf : (float)((double)v * pre_sd);    // snpCaller.py:93-96
                Xp[((h + 2) * 45 + (w + 2)) * 5 + c] = v;
            }
        }
    };
    int64_t site = blockIdx.x;
    if (site < n_sites) { prefetch(site); commit(); }
    __syncthreads();
    for (; site < n_sites; site += gridDim.x) {
        const int64_t nxt = site + gridDim.x;
        if (nxt < n_sites) prefetch(nxt);
        // conv1: 13 tiles of 16 positions; wave w owns tiles w, w+4, w+8 (and 12 for wave 0)
        f12_conv1_pass<2>(Xp, A1, w1r, b1, wv, lane);
        if (wv == 0) f12_conv1_pass<2>(Xp, A1, w1r, b1, 8, lane);
        else f12_conv1_pass<1>(Xp, A1, w1r, b1, 8 + wv, lane);
        __syncthreads();
        // the padded input is free again: write the next site's image while conv2 runs out of A1
        if (nxt < n_sites) commit();
        if (wv < 2) f12_conv2<3>(A1, w2r, b2, A2, wv, lane);
        else f12_conv2<2>(A1, w2r, b2, A2, wv, lane);
        __syncthreads();
        // conv3 reads A2; the next iteration's conv1 only touches Xp / A1, and A2 is not rewritten before the barrier
        // that follows that conv1, so no third barrier is needed
        f12_conv3(A2, w3p, b3, a3 + site * (27 * 64), wv, lane);
    }
}


// =====================================================================================================================
// fp16x3 trunk: the same persistent, weight-stationary fused conv1+conv2+conv3 kernel on the 16x-rate matrix pipe.
// Every fp32 operand v is split as v = hi + lo with hi = fp16(v), lo = fp16(v - hi) (22 significand bits) and a product
// is evaluated as hi*hi + hi*lo + lo*hi with fp32 accumulation in v_mfma_f32_16x16x32_f16 (the dropped lo*lo term is
// 2^-22 relative).  Weights are multiplied by a power of two S on the host before splitting, so their low parts stay
// normal fp16 numbers; accumulators carry S*value and are rescaled (exactly) in the epilogue.  Emulated on the CPU
// with the real weights this is indistinguishable from fp32 rounding (max |dp| 9e-7 vs 9e-7 for plain fp32);
// activations peak near 1.5e3 for every model of the zoo (fp16 max 65504), and are clamped for safety.
//
// LDS layout ("chunk planar", tools/trunk_layout.py checks every access pattern against the gfx950 bank model): all
// operands live in 16-byte SLOTS of 8 halves and every MFMA operand fragment is ONE ds_read_b128 per lane.
//   X   [9 rows x 57] slots, slot = one pixel of the zero-padded 9x45 input.  Two planes with the hi/lo parts of the 5
//       channels interleaved so that the three split products of a tap take TWO MFMAs:
//         XA = [h0 h1 h2 h3 h4 l0 l1 l2]  x  WA = [H0 H1 H2 H3 H4 H0 H1 H2]
//         XB = [l3 l4 h0 h1 h2 h3 h4  0]  x  WB = [H3 H4 L0 L1 L2 L3 L4  0]     (h/l: input, H/L: weight)
//       A K group = 4 taps (lane group g reads the pixel of tap g).  The 25 taps of the 5x5 kernel form 7 groups whose
//       lane-group pairs (g0,g1)/(g2,g3) are horizontally adjacent taps or taps 112 slots apart (row pitch 57 = 41+16),
//       the two cases ds_read_b128's 16-lane groups serve without bank conflicts; the 1x5 and 5x1 kernels reuse the
//       fragments of the groups that contain their taps (2 + 3 groups, zero weights elsewhere): 24 MFMAs per 16 positions.
//   A1  [6 chunks][221 slots]: chunk = 8 of the 48 conv1 channels, slot = h*44 + w.  conv2 (stride 2 in w) reads it
//       conflict-free with tiles (row y, x = 0..15) x 4 + one tile of the 16 left-over columns; chunk pitch is odd.
//   A2  [4 chunks][119 slots], slot = y*26 + x; conv3's lane -> position map is a table (C3_SLOT/C3_OUT).
// =====================================================================================================================
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
constexpr int T_RX = 57, T_XS = 9 * T_RX;                // X slots per padded row / per plane
constexpr int T_R1 = 44, T_PL1 = 221;                     // A1 slots per row / per chunk plane (6 chunks)
constexpr int T_R2 = 26, T_PL2 = 119;                     // A2 slots per row / per chunk plane (4 chunks)
constexpr int T_NW1 = 24, T_NW2 = 18, T_NW3 = 24;         // fragments: conv1 (7+7 5x5, 2+2 1x5, 3+3 5x1), conv2 (9 x 2), conv3 (6 x 4)
constexpr int T_FRAG = 64 * 8;                            // halves per fragment
constexpr int T_XPLANE = T_XS * 8, T_A1PLANE = 6 * T_PL1 * 8, T_A2PLANE = 4 * T_PL2 * 8;   // halves between the two planes of a buffer
// packed blob: [conv1 24 frags][conv2 hi 18][conv2 lo 18][conv3 hi 24][conv3 lo 24] halves, then f32 b1*S[48] b2*S[32]
// b3*S[64] 1/S pad[3], then int32 C3_SLOT[32], C3_OUT[32]
constexpr int H_PACKED_BYTES = 2 * T_FRAG * (T_NW1 + 2 * T_NW2 + 2 * T_NW3) + 4 * (48 + 32 + 64 + 4) + 4 * 64;
// conv1 K groups: tap (dy, dx) of lane group g, as the slot offset dy * T_RX + dx (tools/trunk_layout.py)
constexpr int C1_TAPS[7][4][2] = {{{2, 0}, {2, 1}, {2, 3}, {2, 4}}, {{0, 4}, {2, 2}, {1, 4}, {3, 2}}, {{0, 2}, {0, 3}, {1, 2}, {1, 3}},
                                  {{4, 2}, {4, 3}, {4, 4}, {4, 4}}, {{0, 0}, {0, 1}, {1, 0}, {1, 1}}, {{3, 0}, {3, 1}, {3, 3}, {3, 4}},
                                  {{4, 0}, {4, 1}, {4, 0}, {4, 1}}};
constexpr int C3_SLOT[32] = {60, 34, 32, 8, 28, 26, 68, 16, 2, 54, 14, 56, 6, 62, 52, 58, 30, 36, 34, 6, 4, 66, 38, 0, 66, 42, 8, 12, 10, 40, 64, 34};
constexpr int C3_OUT[32] = {22, 13, 12, 4, 10, 9, 26, 8, 1, 19, 7, 20, 3, 23, 18, 21, 11, 14, -1, -1, 2, 25, 15, 0, -1, 17, -1, 6, 5, 16, 24, -1};

__device__ __forceinline__ h8 as_h8(uint4 v) { union { uint4 u; h8 h; } c; c.u = v; return c.h; }
// four 16-bit fields (one per lane group g): offset in halves of the operand slot of K group G
__host__ __device__ constexpr uint64_t c1_tap_pack(int G)
{
    uint64_t v = 0;
    for (int q = 0; q < 4; q++) v |= (uint64_t)((C1_TAPS[G][q][0] * T_RX + C1_TAPS[G][q][1]) * 8) << (16 * q);
    return v;
}
__host__ __device__ constexpr uint64_t c2_off_pack(int G)
{
    uint64_t v = 0;
    for (int q = 0; q < 4; q++) {
        const int idx = 4 * G + q, tap = idx / 6, ch = idx - tap * 6;            // K chunk = (tap, chunk of 8 channels)
        v |= (uint64_t)((ch * T_PL1 + (tap / 3) * T_R1 + (tap % 3)) * 8) << (16 * q);
    }
    return v;
}
__device__ __forceinline__ h8 lds_h8(const _Float16 *p) { return *reinterpret_cast<const h8 *>(p); }
// Epilogue constants: accumulators hold S * (conv + bias);
//   selu(a / S) = L * max(a, 0) / S + L*A * (exp(min(a, 0) / S) - 1)
// c1 = log2(e) / S, c2 = L / S, c3 = clamp of max(a, 0) that keeps the result inside fp16 range.
struct h_epi { float c1, c2, c3; };
// exp2 with the VOP3 clamp modifier (result clamped to [0,1]): clamp01(exp2(x)) == exp2(min(x, 0)), one instruction
__device__ __forceinline__ float exp2_clamp01(float x) { float r; asm("v_exp_f32_e64 %0, %1 clamp" : "=v"(r) : "v"(x)); return r; }
// v - float(lo / hi half of a packed f16 pair): v_fma_mix_f32 reads the f16 operand directly (no separate v_cvt_f32_f16)
__device__ __forceinline__ float sub_h_lo(float v, uint32_t hpk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v)); return r; }
__device__ __forceinline__ float sub_h_hi(float v, uint32_t hpk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v)); return r; }
#ifndef NC_EPI_VAR
#define NC_EPI_VAR 0
#endif
__device__ __forceinline__ float fma_plain(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ f32x4v selu4_scaled(const f32x4v &acc, const h_epi &k)
{
    f32x4v s;
#ifdef NC_ABL_NOEPI
    return acc;
#endif
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float a = acc[r];
        const float e = exp2_clamp01(a * k.c1);                       // exp(min(a, 0) / S)
        const float pos = __builtin_amdgcn_fmed3f(a, 0.0f, k.c3);
#if NC_EPI_VAR >= 1
        // plain v_fma_f32 (the compiler would pair these into v_pk_fma_f32, which does not run beside MFMAs: tools/ubench/coexec3.hip, epi.hip)
        const float neg = fma_plain(e, SELU_LA, -SELU_LA);
        s[r] = fma_plain(pos, k.c2, neg);
#else
        const float neg = fmaf(e, SELU_LA, -SELU_LA);                 // exactly 0 for a >= 0
        s[r] = fmaf(pos, k.c2, neg);
#endif
    }
    return s;
}
// The MFMAs are issued with the weights as the A operand, so a lane's four accumulator registers are four CONSECUTIVE
// channels (4g .. 4g+3) of ONE position (c16): hi and lo halves go out as one ds_write_b64 each, no cross-lane traffic.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4_store(const f32x4v &v, _Float16 *hp, _Float16 *lp)
{
    const h2 h01 = __builtin_convertvector((f32x2v){v[0], v[1]}, h2), h23 = __builtin_convertvector((f32x2v){v[2], v[3]}, h2);   // v_cvt_pk_f16_f32, RNE
    const uint32_t u01 = __builtin_bit_cast(uint32_t, h01), u23 = __builtin_bit_cast(uint32_t, h23);
#if NC_EPI_VAR >= 2
    // lo = f16(v - f32(hi)) in ONE instruction per value (v_fma_mixlo_f16 / v_fma_mixhi_f16: the fp32 difference is exact, the result is rounded to
    // fp16 like v_cvt_pk_f16_f32 rounds it) instead of v_fma_mix_f32 + half a v_cvt_pk_f16_f32
    uint32_t w01, w23;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(w01) : "v"(u01), "v"(v[0]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(w01) : "v"(u01), "v"(v[1]));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(w23) : "v"(u23), "v"(v[2]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(w23) : "v"(u23), "v"(v[3]));
    *reinterpret_cast<uint2 *>(hp) = make_uint2(u01, u23);
    *reinterpret_cast<uint2 *>(lp) = make_uint2(w01, w23);
#else
    const f32x2v d01 = {sub_h_lo(v[0], u01), sub_h_hi(v[1], u01)}, d23 = {sub_h_lo(v[2], u23), sub_h_hi(v[3], u23)};              // exact in fp32
    const h2 l01 = __builtin_convertvector(d01, h2), l23 = __builtin_convertvector(d23, h2);
    *reinterpret_cast<uint2 *>(hp) = make_uint2(u01, u23);
    *reinterpret_cast<uint2 *>(lp) = make_uint2(__builtin_bit_cast(uint32_t, l01), __builtin_bit_cast(uint32_t, l23));
#endif
}
#ifdef NC_ABL_NOMFMA
#define NC_MFMA(ACC, W, X) asm volatile("" ::"v"(W), "v"(X));
#else
// D[channel 4g + r][position c16] += W[channel][k] * X[k][position]
#ifdef NC_MFMA_NOP
#define NC_MFMA(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(W, X, ACC, 0, 0, 0); asm volatile("s_nop %1" : "+v"(ACC) : "n"(NC_MFMA_NOP));
#else
#define NC_MFMA(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(W, X, ACC, 0, 0, 0);
#endif
#endif
#define NC_MFMA3(ACC, XH, XL, WH, WL) NC_MFMA(ACC, WH, XH) NC_MFMA(ACC, WH, XL) NC_MFMA(ACC, WL, XH)

// conv1 of NT tiles of 16 positions: tile_first, tile_first + 4, ... and, when tile_last >= 0, tile_last as the last one.
// w1 = the 24 resident fragments:
// [0..6] 5x5 WA, [7..13] 5x5 WB, [14,15] 1x5 WA (groups 0,1), [16,17] 1x5 WB, [18..20] 5x1 WA (groups 1,2,3), [21..23] 5x1 WB
// KL: which kernels' output channels the LAST tile computes -- bit 0 the 1x5 kernel's 16 (4 MFMAs per tile, K groups 0,1), bit 1
// the 5x1 kernel's (6 MFMAs, groups 1..3), bit 2 the 5x5 kernel's (14 MFMAs, all groups).  A tile can so be shared by two
// waves to even out the SIMDs, and its MFMAs are interleaved with those of the wave's full tiles (a partial tile on its own
// is one chain of dependent MFMAs: latency-bound).
#ifdef NC_TRACE_P3
#define P3_T(ev) if (trk) trk[ev] = __builtin_readcyclecounter();
#else
#define P3_T(ev)
#endif
template <int NT, int KL = 7, int DP = 1>
__device__ __forceinline__ void t_conv1(const _Float16 *XA, _Float16 *A1H, const h8 (&w1)[T_NW1],
                                        const float *__restrict__ b1s, const h_epi &epi, int tile_first, int lane, int tile_last = -1, int tile_stride = 4, unsigned long long *trk = nullptr, int tev = 0)
{
    // kernel mask and K-group range of tile tm
#define C1_KM(tm) ((tm) == NT - 1 ? KL : 7)
#define C1_GHI(tm) ((C1_KM(tm) & 4) ? 7 : ((C1_KM(tm) & 2) ? 4 : 2))
    const int g = lane >> 4, c16 = lane & 15;
    int xbase[NT], obase[NT];
#pragma unroll
    for (int tm = 0; tm < NT; tm++) {
        const int p = ((tm == NT - 1 && tile_last >= 0) ? tile_last : tile_first + tile_stride * tm) * 16 + c16;
        const int pr = p < 205 ? p : 204;
        const int h = pr / 41, w = pr - h * 41;
        xbase[tm] = (h * T_RX + w) * 8;                              // halves; tap (dy,dx) of pixel (h,w) is padded pixel (h+dy, w+dx)
        // positions 205..207 (tile 12) go to the three unused slots at the end of row 4
        obase[tm] = ((g >> 1) * T_PL1 + (p < 205 ? h * T_R1 + w : 4 * T_R1 + 41 + (p - 205))) * 8 + (g & 1) * 4;
        // opaque per call: keeps site-loop invariant address arithmetic from being hoisted out of the site loop (spills)
        asm volatile("" : "+v"(xbase[tm]), "+v"(obase[tm]));
    }
    const int sh = 16 * g;                                             // field of this lane group in the packed offset constants
    f32x4v acc1[NT], acc2[NT], acc3[NT];
    {
        const f32x4v x1 = *reinterpret_cast<const f32x4v *>(b1s + 4 * g), x2 = *reinterpret_cast<const f32x4v *>(b1s + 16 + 4 * g),
                     x3 = *reinterpret_cast<const f32x4v *>(b1s + 32 + 4 * g);
#pragma unroll
        for (int tm = 0; tm < NT; tm++) { acc1[tm] = x1; acc2[tm] = x2; acc3[tm] = x3; }
    }
    // software pipeline: the ds_reads of group G + 1 are issued before the MFMAs of group G (register double buffer)
    constexpr int NB = DP + 1;
    h8 xa[NB][NT], xb[NB][NT];
#ifdef NC_ABL_LDSDUMMY
    h8 dm[NB][NT][2];
#endif
    auto load1 = [&](int G, int slot) {
        const int toff = (int)((c1_tap_pack(G) >> sh) & 0xffffu);       // halves
#pragma unroll
        for (int tm = 0; tm < NT; tm++) {
            if (G >= C1_GHI(tm)) continue;
#ifdef NC_ABL_LDSDUMMY
            #ifdef NC_ABL_LDSLINEAR
            { h8 d0 = lds_h8(XA + lane * 8 + (G * NT + tm) * 512), d1 = lds_h8(XA + lane * 8 + (G * NT + tm) * 512 + T_XPLANE); dm[slot][tm][0] = d0; dm[slot][tm][1] = d1; }
#else
            { h8 d0 = lds_h8(XA + xbase[tm] + toff), d1 = lds_h8(XA + xbase[tm] + toff + T_XPLANE); dm[slot][tm][0] = d0; dm[slot][tm][1] = d1; }
#endif
#endif
#if defined(NC_ABL_NOLDS) || defined(NC_ABL_LDSDUMMY)
            xa[slot][tm] = w1[(G + tm) % 7]; xb[slot][tm] = w1[7 + (G + 2 * tm) % 7];
#else
            xa[slot][tm] = lds_h8(XA + xbase[tm] + toff);
            xb[slot][tm] = lds_h8(XA + xbase[tm] + toff + T_XPLANE);
#endif
        }
    };
    constexpr int NG = NT > 1 ? 7 : C1_GHI(0);
#pragma unroll
    for (int G = 0; G < DP && G < NG; G++) load1(G, G % NB);
#pragma unroll
    for (int G = 0; G < NG; G++) {
        const int cur = G % NB;
        if (G + DP < NG) load1(G + DP, (G + DP) % NB);
        __builtin_amdgcn_sched_barrier(0);
        // independent accumulators interleaved: no MFMA depends on the one issued just before it
#pragma unroll
        for (int tm = 0; tm < NT; tm++) if (C1_KM(tm) & 4) { NC_MFMA(acc3[tm], w1[G], xa[cur][tm]) }
        if (G < 2) {
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (C1_KM(tm) & 1) { NC_MFMA(acc1[tm], w1[14 + G], xa[cur][tm]) }
        }
        if (G >= 1 && G <= 3) {
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (C1_KM(tm) & 2) { NC_MFMA(acc2[tm], w1[18 + G - 1], xa[cur][tm]) }
        }
#pragma unroll
        for (int tm = 0; tm < NT; tm++) if (C1_KM(tm) & 4) { NC_MFMA(acc3[tm], w1[7 + G], xb[cur][tm]) }
        if (G < 2) {
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (C1_KM(tm) & 1) { NC_MFMA(acc1[tm], w1[16 + G], xb[cur][tm]) }
        }
        if (G >= 1 && G <= 3) {
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (C1_KM(tm) & 2) { NC_MFMA(acc2[tm], w1[21 + G - 1], xb[cur][tm]) }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef NC_ABL_LDSDUMMY
#pragma unroll
        for (int tm = 0; tm < NT; tm++) if (G < C1_GHI(tm)) asm volatile("" ::"v"(dm[cur][tm][0]), "v"(dm[cur][tm][1]));
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    if (trk) trk[tev] = __builtin_readcyclecounter();
#pragma unroll
    for (int tm = 0; tm < NT; tm++) {
        const int o = obase[tm];
        if (C1_KM(tm) & 1) split4_store(selu4_scaled(acc1[tm], epi), A1H + o, A1H + o + T_A1PLANE);
        if (C1_KM(tm) & 2) split4_store(selu4_scaled(acc2[tm], epi), A1H + o + 2 * T_PL1 * 8, A1H + o + 2 * T_PL1 * 8 + T_A1PLANE);
        if (C1_KM(tm) & 4) split4_store(selu4_scaled(acc3[tm], epi), A1H + o + 4 * T_PL1 * 8, A1H + o + 4 * T_PL1 * 8 + T_A1PLANE);
    }
#undef C1_KM
#undef C1_GHI
}

// One full conv1 tile in two halves: c1_tile_load requests the operands of all 7 K groups (14 fragments; the site's X buffer is
// complete long before), c1_tile_mma multiplies them.  Role C issues the first before the beta barrier and runs the second behind it:
// a single tile is one chain of {request, wait, 2-4 MFMAs} steps otherwise, bound by the LDS round trip of every step.
__device__ __forceinline__ void c1_tile_load(const _Float16 *XA, int tile, int lane, h8 (&xa)[7], h8 (&xb)[7])
{
    const int g = lane >> 4, c16 = lane & 15, sh = 16 * g;
    const int p = tile * 16 + c16, pr = p < 205 ? p : 204, h = pr / 41, w = pr - h * 41;
    int xbase = (h * T_RX + w) * 8;
    asm volatile("" : "+v"(xbase));
#pragma unroll
    for (int G = 0; G < 7; G++) {
        const int toff = (int)((c1_tap_pack(G) >> sh) & 0xffffu);
        xa[G] = lds_h8(XA + xbase + toff);
        xb[G] = lds_h8(XA + xbase + toff + T_XPLANE);
    }
}
__device__ __forceinline__ void c1_tile_mma(_Float16 *A1H, const h8 (&w1)[T_NW1], const float *__restrict__ b1s, const h_epi &epi, int tile, int lane,
                                            const h8 (&xa)[7], const h8 (&xb)[7])
{
    const int g = lane >> 4, c16 = lane & 15;
    const int p = tile * 16 + c16, pr = p < 205 ? p : 204, h = pr / 41, w = pr - h * 41;
    int o = ((g >> 1) * T_PL1 + (p < 205 ? h * T_R1 + w : 4 * T_R1 + 41 + (p - 205))) * 8 + (g & 1) * 4;
    asm volatile("" : "+v"(o));
    f32x4v acc1 = *reinterpret_cast<const f32x4v *>(b1s + 4 * g), acc2 = *reinterpret_cast<const f32x4v *>(b1s + 16 + 4 * g),
           acc3 = *reinterpret_cast<const f32x4v *>(b1s + 32 + 4 * g);
#pragma unroll
    for (int G = 0; G < 7; G++) {
        NC_MFMA(acc3, w1[G], xa[G])
        if (G < 2) { NC_MFMA(acc1, w1[14 + G], xa[G]) }
        if (G >= 1 && G <= 3) { NC_MFMA(acc2, w1[18 + G - 1], xa[G]) }
        NC_MFMA(acc3, w1[7 + G], xb[G])
        if (G < 2) { NC_MFMA(acc1, w1[16 + G], xb[G]) }
        if (G >= 1 && G <= 3) { NC_MFMA(acc2, w1[21 + G - 1], xb[G]) }
    }
    split4_store(selu4_scaled(acc1, epi), A1H + o, A1H + o + T_A1PLANE);
    split4_store(selu4_scaled(acc2, epi), A1H + o + 2 * T_PL1 * 8, A1H + o + 2 * T_PL1 * 8 + T_A1PLANE);
    split4_store(selu4_scaled(acc3, epi), A1H + o + 4 * T_PL1 * 8, A1H + o + 4 * T_PL1 * 8 + T_A1PLANE);
}

// conv2: wave (tn = wv & 1, t0 = wv >> 1) computes output channels 16 tn .. 16 tn + 15 of tiles t0, t0 + 2, (t0 + 4).
// Tiles 0..3 = output row y, x = 0..15; tile 4 = columns 16..19 of the four rows (lane c16 -> y = c16 & 3, x = 16 + c16/4).
template <int NT>
__device__ __forceinline__ void t_conv2(const _Float16 *A1H, _Float16 *A2H, const h8 (&wh)[9],
                                        const h8 (&wl)[9], const float *__restrict__ b2s, const h_epi &epi, int tn, int t0, int lane)
{
    const int g = lane >> 4, c16 = lane & 15;
    int abase[NT], obase[NT];
#pragma unroll
    for (int tm = 0; tm < NT; tm++) {
        const int t = t0 + 2 * tm;
        const int y = t < 4 ? t : (c16 & 3), x = t < 4 ? c16 : 16 + (c16 >> 2);
        abase[tm] = (y * T_R1 + 2 * x) * 8;
        obase[tm] = ((2 * tn + (g >> 1)) * T_PL2 + y * T_R2 + x) * 8 + (g & 1) * 4;
        asm volatile("" : "+v"(abase[tm]), "+v"(obase[tm]));
    }
    const int sh = 16 * g;
    f32x4v acc[NT];
    {
        const f32x4v b = *reinterpret_cast<const f32x4v *>(b2s + tn * 16 + 4 * g);
#pragma unroll
        for (int tm = 0; tm < NT; tm++) acc[tm] = b;
    }
#ifndef NC_C2_DEPTH_L
#define NC_C2_DEPTH_L 1
#endif
#ifndef NC_C2_DEPTH_H
#define NC_C2_DEPTH_H 1
#endif
    constexpr int DP = NT == 2 ? NC_C2_DEPTH_L : NC_C2_DEPTH_H, NB = DP + 1;
    h8 ah[NB][NT], al[NB][NT];
#ifdef NC_ABL_LDSDUMMY
    h8 dm[NB][NT][2];
#endif
    auto load2 = [&](int G, int slot) {
        const int off = (int)((c2_off_pack(G) >> sh) & 0xffffu);
#pragma unroll
        for (int tm = 0; tm < NT; tm++) {
#ifdef NC_ABL_LDSDUMMY
#ifdef NC_ABL_LDSLINEAR
            dm[slot][tm][0] = lds_h8(A1H + lane * 8 + (G * NT + tm) * 512); dm[slot][tm][1] = lds_h8(A1H + lane * 8 + (G * NT + tm) * 512 + T_A1PLANE);
#else
            dm[slot][tm][0] = lds_h8(A1H + abase[tm] + off); dm[slot][tm][1] = lds_h8(A1H + abase[tm] + off + T_A1PLANE);
#endif
#endif
#if defined(NC_ABL_NOLDS) || defined(NC_ABL_LDSDUMMY)
            ah[slot][tm] = wh[(G + tm) % 9]; al[slot][tm] = wl[(G + 2 * tm) % 9];
#else
            ah[slot][tm] = lds_h8(A1H + abase[tm] + off);
            al[slot][tm] = lds_h8(A1H + abase[tm] + off + T_A1PLANE);
#endif
        }
    };
#pragma unroll
    for (int G = 0; G < DP; G++) load2(G, G % NB);
#pragma unroll
    for (int G = 0; G < 9; G++) {
        const int cur = G % NB;
        if (G + DP < 9) load2(G + DP, (G + DP) % NB);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < NT; tm++) { NC_MFMA(acc[tm], wh[G], ah[cur][tm]) }
#ifndef NC_EXP_C2_DROP_AL
#pragma unroll
        for (int tm = 0; tm < NT; tm++) { NC_MFMA(acc[tm], wh[G], al[cur][tm]) }
#endif
#ifndef NC_EXP_C2_DROP_WL
#pragma unroll
        for (int tm = 0; tm < NT; tm++) { NC_MFMA(acc[tm], wl[G], ah[cur][tm]) }
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifdef NC_ABL_LDSDUMMY
#pragma unroll
        for (int tm = 0; tm < NT; tm++) asm volatile("" ::"v"(dm[cur][tm][0]), "v"(dm[cur][tm][1]));
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
#pragma unroll
    for (int tm = 0; tm < NT; tm++) split4_store(selu4_scaled(acc[tm], epi), A2H + obase[tm], A2H + obase[tm] + T_A2PLANE);
}

// conv3: wave wv computes output channels 16 wv .. 16 wv + 15 of both position tiles (lane -> position: C3_SLOT / C3_OUT)
__device__ __forceinline__ void t_conv3(const _Float16 *A2H, const h8 (&w3h)[6], const h8 (&w3l)[6],
                                        const float *__restrict__ b3s, const h_epi &epi, float *__restrict__ out_site,
                                        const int (&c3slot)[2], const int (&c3out)[2], int wv, int lane)
{
    const int g = lane >> 4;
    int abase[2];
#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
        abase[tm] = (g * T_PL2 + c3slot[tm]) * 8;
        asm volatile("" : "+v"(abase[tm]));
    }
    f32x4v acc[2];
    acc[0] = *reinterpret_cast<const f32x4v *>(b3s + wv * 16 + 4 * g);
    acc[1] = acc[0];
#ifndef NC_C3_DEPTH
#define NC_C3_DEPTH 1
#endif
    constexpr int DP = NC_C3_DEPTH, NB = DP + 1;                      // K groups requested ahead of the one being multiplied / register buffers
    h8 ah[NB][2], al[NB][2];
#ifdef NC_ABL_LDSDUMMY
    h8 dm[NB][2][2];
#endif
    auto load3 = [&](int G, int slot) {                   // K group G = tap G, lane group g = channel chunk g
        const int off = ((G / 3) * T_R2 + (G % 3)) * 8;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
#ifdef NC_ABL_LDSDUMMY
#ifdef NC_ABL_LDSLINEAR
            dm[slot][tm][0] = lds_h8(A2H + lane * 8 + (G % 3) * 1024 + tm * 512); dm[slot][tm][1] = lds_h8(A2H + lane * 8 + (G % 3) * 1024 + tm * 512 + T_A2PLANE);
#else
            dm[slot][tm][0] = lds_h8(A2H + abase[tm] + off); dm[slot][tm][1] = lds_h8(A2H + abase[tm] + off + T_A2PLANE);
#endif
#endif
#if defined(NC_ABL_NOLDS) || defined(NC_ABL_LDSDUMMY)
            ah[slot][tm] = w3h[(G + tm) % 6]; al[slot][tm] = w3l[(G + 2 * tm) % 6];
#else
            ah[slot][tm] = lds_h8(A2H + abase[tm] + off);
            al[slot][tm] = lds_h8(A2H + abase[tm] + off + T_A2PLANE);
#endif
        }
    };
#pragma unroll
    for (int G = 0; G < DP; G++) load3(G, G % NB);
#pragma unroll
    for (int G = 0; G < 6; G++) {
        const int cur = G % NB;
        if (G + DP < 6) load3(G + DP, (G + DP) % NB);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 2; tm++) { NC_MFMA(acc[tm], w3h[G], ah[cur][tm]) }
#ifndef NC_EXP_C3_DROP_AL
#pragma unroll
        for (int tm = 0; tm < 2; tm++) { NC_MFMA(acc[tm], w3h[G], al[cur][tm]) }
#endif
#ifndef NC_EXP_C3_DROP_WL
#pragma unroll
        for (int tm = 0; tm < 2; tm++) { NC_MFMA(acc[tm], w3l[G], ah[cur][tm]) }
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifdef NC_ABL_LDSDUMMY
#pragma unroll
        for (int tm = 0; tm < 2; tm++) asm volatile("" ::"v"(dm[cur][tm][0]), "v"(dm[cur][tm][1]));
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    const h_epi &e3 = epi;                                            // same fp16 range clamp as the other layers: k6_fc1_h3 splits
                                                                      // these values into fp16 hi/lo without a clamp of its own
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
        if (c3out[tm] >= 0) *reinterpret_cast<f32x4v *>(out_site + c3out[tm] * 64 + wv * 16 + 4 * g) = selu4_scaled(acc[tm], e3);
}

// Wave-specialised persistent kernel: one 512-thread workgroup per CU, two waves per SIMD with 256 VGPRs each.
//   waves 0-3 ("C"): conv1.  The 24 conv1 weight fragments (96 VGPRs) stay in registers for the kernel's lifetime; the waves
//                    also stage the next site's input tensor (thread = pixel) into the other X buffer.
//   waves 4-7 ("D"): conv2 + conv3.  This wave's 9+9 conv2 and 6+6 conv3 fragments (120 VGPRs) stay in registers.
// C works on site k+1 while D works on site k: X and A1 are double-buffered, A2 is single.  Two workgroup barriers per
// site (alpha_k: A1[k&1] and X[(k+1)&1] complete; beta_k: A2 complete), executed by both roles in the same order:
//   C:  P0 | conv1(0) alpha_0 | conv1(1) first tiles, staging commit of site 2, beta_0, last tile, alpha_1 | ... | beta_last
//   D:  P0 | alpha_0 conv2(0) beta_0 conv3(0) | alpha_1 conv2(1) beta_1 conv3(1) | ...
// No weight is re-read per site, and the MFMA phases of one role overlap the epilogues of the other on every SIMD.
#if defined(NC_TRACE) || defined(NC_TRACE_BLOCKS) || defined(NC_TRACE_P3)
__device__ unsigned long long nc_trace_buf[8][8][8];     // [wave][site k in 8..15][event]
#endif
#if defined(NC_TRACE) && !defined(NC_TRACE_BLOCKS)
#define NC_T(ev) if (blockIdx.x == 3 && lane == 0 && k >= 8 && k < 16) { nc_trace_buf[wv][k - 8][ev] = __builtin_readcyclecounter(); if (ev == 0) nc_trace_buf[wv][k - 8][7] = __builtin_amdgcn_s_memrealtime(); }
#else
#define NC_T(ev)
#endif
#ifdef NC_ABL_NOBAR
#define NC_SITE_SYNC()
#else
#define NC_SITE_SYNC() __syncthreads()
#endif
template <bool X16>                                             // X16: the site tensors are int16 (nc_set_tensor_format(ctx, 1))
__global__ __launch_bounds__(512) void k5_trunk_h3(const float *__restrict__ x, const uint8_t *__restrict__ wp, float *__restrict__ a3,
                                                   int64_t n_sites, const double *__restrict__ scale, int scale_mode, int64_t site0, float x_limit,
                                                   uint8_t *__restrict__ range_sites)
{
    // [buffer][plane]: the second plane of a buffer sits at a constant distance (< 64 KB) from the first, so one address
    // VGPR + the DS instruction's immediate offset serves both
    __shared__ __attribute__((aligned(16))) _Float16 X[2][2 * T_XPLANE];
    __shared__ __attribute__((aligned(16))) _Float16 A1[2][2 * T_A1PLANE];
    __shared__ __attribute__((aligned(16))) _Float16 A2[2 * T_A2PLANE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#ifdef NC_TRACE_BLOCKS
    if (threadIdx.x == 0 && blockIdx.x < 128) (&nc_trace_buf[0][0][0])[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef NC_TRACE_BLOCKS
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 3) (&nc_trace_buf[0][0][0])[504 + (threadIdx.x >> 6)] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | (1ull << 40);
#endif
    const uint4 *w1f = reinterpret_cast<const uint4 *>(wp), *w2h = w1f + T_NW1 * 64, *w2l = w2h + T_NW2 * 64, *w3h = w2l + T_NW2 * 64,
                *w3l = w3h + T_NW3 * 64;
    // the scaled biases live in LDS: a global load inside the site loop would make its s_waitcnt vmcnt also wait for the
    // staging loads (role C) / the activation stores (role D) issued before it
    __shared__ __attribute__((aligned(16))) float BIAS[48 + 32 + 64];
    const float *bg = reinterpret_cast<const float *>(w3l + T_NW3 * 64);
    const float *b1s = BIAS, *b2s = BIAS + 48, *b3s = BIAS + 80;
    const int *c3tab = reinterpret_cast<const int *>(bg + 48 + 32 + 68);
    const float inv_s = bg[48 + 32 + 64];
    if (threadIdx.x < 48 + 32 + 64) BIAS[threadIdx.x] = bg[threadIdx.x];
    const h_epi epi = {inv_s * 1.44269504088896341f, inv_s * SELU_L, 60000.0f / (inv_s * SELU_L)};
    const int64_t n_k = (n_sites - blockIdx.x + gridDim.x - 1) / gridDim.x;        // sites of this workgroup (>= 1)
    for (int i = threadIdx.x; i < 4 * T_XS; i += 512) *reinterpret_cast<uint4 *>(&X[0][0] + i * 8) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (wv < 4) {
        // ------------------------------------------------------------------ role C: staging + conv1
        h8 w1[T_NW1];
#pragma unroll
        for (int q = 0; q < T_NW1; q++) w1[q] = as_h8(w1f[q * 64 + lane]);
        // staging: thread t < 205 owns pixel t = h*41 + w (5 channels = 20 contiguous bytes of the site's tensor)
        const int px = threadIdx.x < 205 ? threadIdx.x : 204, ph = px / 41, pw = px - ph * 41;
        const int xslot = ((ph + 2) * T_RX + pw + 2) * 8;
        // prefetch() only ISSUES the loads (raw bits stay in registers); every conversion happens in commit(), a conv1 call
        // later, so that no wave waits for memory at the top of a site
        float pre[5];
        uint32_t raw[3];
        double pre_sd = 1.0;
        int64_t pre_site = 0;
        auto prefetch = [&](int64_t site) {
            pre_site = site;
            if constexpr (X16) {
                const int16_t *xs = reinterpret_cast<const int16_t *>(x) + site * NC_SNP_TENSOR + px * 5;       // 2-byte aligned
                typedef uint32_t __attribute__((aligned(2))) u32_a2;
                raw[0] = *reinterpret_cast<const u32_a2 *>(xs);
                raw[1] = *reinterpret_cast<const u32_a2 *>(xs + 2);
                raw[2] = (uint32_t)(uint16_t)xs[4];
            } else {
                const float *xs = x + site * NC_SNP_TENSOR + px * 5;
#pragma unroll
                for (int u = 0; u < 5; u++) pre[u] = xs[u];
            }
            if (scale) pre_sd = scale[site0 + site];                                  // consumed in commit(): no wait here
        };
        auto commit = [&](int buf) {
            if (threadIdx.x < 205) {
                if constexpr (X16) {
                    pre[0] = (float)(int16_t)(raw[0] & 0xffffu); pre[1] = (float)(int16_t)(raw[0] >> 16);
                    pre[2] = (float)(int16_t)(raw[1] & 0xffffu); pre[3] = (float)(int16_t)(raw[1] >> 16);
                    pre[4] = (float)(int16_t)raw[2];
                }
                // snpCaller.py:93-96: rows 1..4, channels 0..3 are scaled; a multiplier of exactly 1 elsewhere keeps this branch-free
                const double md = (scale && ph > 0) ? pre_sd : 1.0;
                const float mf = (float)md;
                if (scale_mode == 0) {
#pragma unroll
                    for (int u = 0; u < 4; u++) pre[u] *= mf;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++) pre[u] = (float)((double)pre[u] * md);
                }
                // range guard: the epilogues clamp activations to the fp16 range; below x_limit the weights' L1 norms prove that none gets
                // there (nc_load_weights), above it the site is flagged and the caller re-runs it on the exact fp32 trunk
                const float amax = fmaxf(fmaxf(fmaxf(fabsf(pre[0]), fabsf(pre[1])), fmaxf(fabsf(pre[2]), fabsf(pre[3]))), fabsf(pre[4]));
                if (range_sites && !(amax <= x_limit)) range_sites[site0 + pre_site] = 1;
                _Float16 hi[5], lo[5];
#pragma unroll
                for (int u = 0; u < 5; u++) {
                    float v = pre[u];
                    v = fminf(fmaxf(v, -65504.0f), 65504.0f);
                    hi[u] = (_Float16)v;
                    lo[u] = (_Float16)(v - (float)hi[u]);
                }
                const h8 sa = {hi[0], hi[1], hi[2], hi[3], hi[4], lo[0], lo[1], lo[2]};
                const h8 sb = {lo[3], lo[4], hi[0], hi[1], hi[2], hi[3], hi[4], (_Float16)0.0f};
                *reinterpret_cast<h8 *>(&X[buf][xslot]) = sa;
                *reinterpret_cast<h8 *>(&X[buf][xslot + T_XPLANE]) = sb;
            }
        };
        int64_t site = blockIdx.x;
        prefetch(site);
        commit(0);
        __syncthreads();                                                           // P0
#ifdef NC_TRACE_BLOCKS
        if (threadIdx.x == 0 && blockIdx.x < 128) (&nc_trace_buf[0][0][0])[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
#endif
        for (int64_t k = 0; k < n_k; k++, site += gridDim.x) {
            const int buf = (int)(k & 1);
            const bool more = k + 1 < n_k;
#ifndef NC_ABL_NOSTAGE
            if (more) prefetch(site + gridDim.x);
#endif
            NC_T(0)
#ifndef NC_ABL_NOC
            // 13 tiles over 4 waves, 9 of them before beta: wave w its tiles w and w + 4; tile 8 is shared, wave 0 computes its
            // 5x5 channels (14 MFMAs) and wave 1 the 1x5 + 5x1 ones (10), interleaved with their full tiles
            if (wv == 0) t_conv1<3, 4>(X[buf], A1[buf], w1, b1s, epi, wv, lane, 8);
            else if (wv == 1) t_conv1<3, 3>(X[buf], A1[buf], w1, b1s, epi, wv, lane, 8);
            else t_conv1<2>(X[buf], A1[buf], w1, b1s, epi, wv, lane);
#endif
            NC_T(1)
            // the other X buffer's last reader was conv1 of site k-1 (finished before alpha_{k-1}): the next site's tensor is
            // committed here, where role C has slack, and long after its loads were issued
#if defined(NC_C1_PRELOAD) && !defined(NC_ABL_NOC)
            h8 lxa[7], lxb[7];
            c1_tile_load(X[buf], wv == 0 ? 12 : 8 + wv, lane, lxa, lxb);
#endif
#ifndef NC_ABL_NOSTAGE
            if (more) commit(buf ^ 1);
#endif
            NC_T(6)
            if (k > 0) NC_SITE_SYNC();                                            // beta_{k-1}
            NC_T(2)
#ifndef NC_ABL_NOC
#ifdef NC_C1_PRELOAD
            c1_tile_mma(A1[buf], w1, b1s, epi, wv == 0 ? 12 : 8 + wv, lane, lxa, lxb);
#else
#ifndef NC_C1_LAST_DEPTH
#define NC_C1_LAST_DEPTH 1
#endif
            t_conv1<1, 7, NC_C1_LAST_DEPTH>(X[buf], A1[buf], w1, b1s, epi, wv == 0 ? 12 : 8 + wv, lane);
#endif
#endif
            NC_T(3)
            NC_T(4)
            NC_SITE_SYNC();                                                       // alpha_k
            NC_T(5)
        }
        __syncthreads();                                                           // beta_{n_k - 1}
#ifdef NC_TRACE_BLOCKS
        if (threadIdx.x == 0 && blockIdx.x < 128) (&nc_trace_buf[0][0][0])[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
#endif
    } else {
        // ------------------------------------------------------------------ role D: conv2 + conv3
        const int d = wv - 4, tn = d & 1, heavy = d >> 1;                          // heavy: conv2 tiles 0,2,4; light: tiles 1,3
        h8 c2h[9], c2l[9], c3h[6], c3l[6];
#pragma unroll
        for (int q = 0; q < 9; q++) { c2h[q] = as_h8(w2h[(q * 2 + tn) * 64 + lane]); c2l[q] = as_h8(w2l[(q * 2 + tn) * 64 + lane]); }
#pragma unroll
        for (int q = 0; q < 6; q++) { c3h[q] = as_h8(w3h[(q * 4 + d) * 64 + lane]); c3l[q] = as_h8(w3l[(q * 4 + d) * 64 + lane]); }
        const int c3slot[2] = {c3tab[lane & 15], c3tab[16 + (lane & 15)]}, c3out[2] = {c3tab[32 + (lane & 15)], c3tab[48 + (lane & 15)]};
        __syncthreads();                                                           // P0
        int64_t site = blockIdx.x;
        for (int64_t k = 0; k < n_k; k++, site += gridDim.x) {
            const int buf = (int)(k & 1);
            NC_T(0)
            NC_SITE_SYNC();                                                       // alpha_k
            NC_T(1)
#ifndef NC_ABL_NOD
            if (heavy) t_conv2<3>(A1[buf], A2, c2h, c2l, b2s, epi, tn, 0, lane);
            else t_conv2<2>(A1[buf], A2, c2h, c2l, b2s, epi, tn, 1, lane);
#endif
            NC_T(2)
            NC_SITE_SYNC();                                                       // beta_k
            NC_T(3)
#ifndef NC_ABL_NOD
            t_conv3(A2, c3h, c3l, b3s, epi, a3 + site * (27 * 64), c3slot, c3out, d, lane);
#endif
            NC_T(4)
        }
    }
}

// ---- the three-stage form of the trunk (k5_trunk_p3): conv1, conv2 and conv3 work on three consecutive sites, ONE workgroup barrier per
// site.  A conv2 wave computes all 32 output channels of its tiles and a conv3 wave 32 of the 64, so that every operand fragment read from
// LDS feeds 6 MFMAs instead of 3 (338 instead of 458 ds_read_b128 per site: the LDS pipe at 128 B/clk is as loaded as the matrix pipe).
// conv2, wave CW of two: tiles CW and CW + 2 in full, and channel half CW of tile 4 (columns 16..19 of the four rows)
#ifdef NC_P3_NOP
#define NC_MFMA_P(ACC, W, X) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(W, X, ACC, 0, 0, 0); asm volatile("s_nop %1" : "+v"(ACC) : "n"(NC_P3_NOP));
#else
#define NC_MFMA_P(ACC, W, X) NC_MFMA(ACC, W, X)
#endif
// the accumulators of a conv2 wave between its MFMA loop and its epilogue (k5_trunk_lin runs the epilogue a step later: the other wave of the SIMD is
// in its MFMA loop then)
struct c2_acc { f32x4v a[2][2], a4; };
template <int CW>
__device__ __forceinline__ void t_conv2_pair_mma(const _Float16 *A1H, const h8 (&wh)[9][2], const h8 (&wl)[9][2], const float *__restrict__ b2s, int lane, c2_acc &o)
{
    const int g = lane >> 4, c16 = lane & 15, sh = 16 * g;
    int abase[3];
#pragma unroll
    for (int tm = 0; tm < 3; tm++) {
        const int t = tm < 2 ? CW + 2 * tm : 4;
        const int y = t < 4 ? t : (c16 & 3), x = t < 4 ? c16 : 16 + (c16 >> 2);
        abase[tm] = (y * T_R1 + 2 * x) * 8;
        asm volatile("" : "+v"(abase[tm]));
    }
    f32x4v acc[2][2], acc4;
    {
        const f32x4v b0 = *reinterpret_cast<const f32x4v *>(b2s + 4 * g), b1 = *reinterpret_cast<const f32x4v *>(b2s + 16 + 4 * g);
        acc[0][0] = b0; acc[1][0] = b0; acc[0][1] = b1; acc[1][1] = b1;
        acc4 = CW ? b1 : b0;
    }
#ifndef NC_P3_C2_DEPTH
#define NC_P3_C2_DEPTH 1
#endif
    constexpr int DP = NC_P3_C2_DEPTH, NB = DP + 1;
    h8 ah[NB][3], al[NB][3];
    auto load2 = [&](int G, int slot) {
        const int off = (int)((c2_off_pack(G) >> sh) & 0xffffu);
#pragma unroll
        for (int tm = 0; tm < 3; tm++) {
            ah[slot][tm] = lds_h8(A1H + abase[tm] + off);
            al[slot][tm] = lds_h8(A1H + abase[tm] + off + T_A1PLANE);
        }
    };
#pragma unroll
    for (int G = 0; G < DP; G++) load2(G, G % NB);
#pragma unroll
    for (int G = 0; G < 9; G++) {
        const int cur = G % NB;
        if (G + DP < 9) load2(G + DP, (G + DP) % NB);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 2; tm++) { NC_MFMA_P(acc[tm][0], wh[G][0], ah[cur][tm]) NC_MFMA_P(acc[tm][1], wh[G][1], ah[cur][tm]) }
        NC_MFMA_P(acc4, wh[G][CW], ah[cur][2])
#pragma unroll
        for (int tm = 0; tm < 2; tm++) { NC_MFMA_P(acc[tm][0], wh[G][0], al[cur][tm]) NC_MFMA_P(acc[tm][1], wh[G][1], al[cur][tm]) }
        NC_MFMA_P(acc4, wh[G][CW], al[cur][2])
#pragma unroll
        for (int tm = 0; tm < 2; tm++) { NC_MFMA_P(acc[tm][0], wl[G][0], ah[cur][tm]) NC_MFMA_P(acc[tm][1], wl[G][1], ah[cur][tm]) }
        NC_MFMA_P(acc4, wl[G][CW], ah[cur][2])
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) o.a[tm][tn] = acc[tm][tn];
    o.a4 = acc4;
}
template <int CW>
__device__ __forceinline__ void t_conv2_pair_epi(_Float16 *A2H, const h_epi &epi, int lane, const c2_acc &o)
{
    const int g = lane >> 4, c16 = lane & 15;
    int obase[3];
#pragma unroll
    for (int tm = 0; tm < 3; tm++) {
        const int t = tm < 2 ? CW + 2 * tm : 4;
        const int y = t < 4 ? t : (c16 & 3), x = t < 4 ? c16 : 16 + (c16 >> 2);
        obase[tm] = ((g >> 1) * T_PL2 + y * T_R2 + x) * 8 + (g & 1) * 4;              // channel half 0; half 1: + 2 * T_PL2 * 8
        asm volatile("" : "+v"(obase[tm]));
    }
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
            const int oo = obase[tm] + tn * 2 * T_PL2 * 8;
            split4_store(selu4_scaled(o.a[tm][tn], epi), A2H + oo, A2H + oo + T_A2PLANE);
        }
    {
        const int oo = obase[2] + CW * 2 * T_PL2 * 8;
        split4_store(selu4_scaled(o.a4, epi), A2H + oo, A2H + oo + T_A2PLANE);
    }
}
// conv2 in two phases inside one step (k5_trunk_lin, NC_LIN_PIPE): phase A = the MFMAs of tile CW and of the wave's half of tile 4; phase B = the
// MFMAs of tile CW + 2 with the EPILOGUE of phase A's three accumulators (SELU, hi / lo split, stores) issued between them -- a wave's own vector
// instructions ride in the shadow of its own MFMAs (about three issue slots per 16-cycle MFMA), which the two waves of a SIMD do not do for
// each other (the issue port goes to the older wave: per-wave phase times, profiles/r06_trunk_phases.md); then the epilogue of tile CW + 2.
// Every operand fragment is still read once and feeds 6 (tile 4: 3) MFMAs.  The interleave is requested from the scheduler per K group:
// 6 x {1 MFMA, NC_PIPE_NV vector instructions}.
#ifndef NC_PIPE_NV
#define NC_PIPE_NV 3
#endif
template <int CW>
__device__ __forceinline__ void t_conv2_two_phase(const _Float16 *A1H, _Float16 *A2H, const h8 (&wh)[9][2], const h8 (&wl)[9][2], const float *__restrict__ b2s,
                                                  const h_epi &epi, int lane, unsigned long long *trk = nullptr)
{
    const int g = lane >> 4, c16 = lane & 15, sh = 16 * g;
    int abase[3], obase[3];
#pragma unroll
    for (int tm = 0; tm < 3; tm++) {
        const int t = tm < 2 ? CW + 2 * tm : 4;
        const int y = t < 4 ? t : (c16 & 3), x = t < 4 ? c16 : 16 + (c16 >> 2);
        abase[tm] = (y * T_R1 + 2 * x) * 8;
        obase[tm] = ((g >> 1) * T_PL2 + y * T_R2 + x) * 8 + (g & 1) * 4;
        asm volatile("" : "+v"(abase[tm]), "+v"(obase[tm]));
    }
    f32x4v acc[2][2], acc4;
    {
        const f32x4v b0 = *reinterpret_cast<const f32x4v *>(b2s + 4 * g), b1 = *reinterpret_cast<const f32x4v *>(b2s + 16 + 4 * g);
        acc[0][0] = b0; acc[1][0] = b0; acc[0][1] = b1; acc[1][1] = b1;
        acc4 = CW ? b1 : b0;
    }
    constexpr int DP = 1, NB = DP + 1;
    h8 ah[NB][2], al[NB][2];
    // ---- phase A: tiles CW (index 0) and 4 (index 2)
    auto loadA = [&](int G, int slot) {
        const int off = (int)((c2_off_pack(G) >> sh) & 0xffffu);
        ah[slot][0] = lds_h8(A1H + abase[0] + off); al[slot][0] = lds_h8(A1H + abase[0] + off + T_A1PLANE);
        ah[slot][1] = lds_h8(A1H + abase[2] + off); al[slot][1] = lds_h8(A1H + abase[2] + off + T_A1PLANE);
    };
    loadA(0, 0);
#pragma unroll
    for (int G = 0; G < 9; G++) {
        const int cur = G % NB;
        if (G + DP < 9) loadA(G + DP, (G + DP) % NB);
        __builtin_amdgcn_sched_barrier(0);
        NC_MFMA_P(acc[0][0], wh[G][0], ah[cur][0]) NC_MFMA_P(acc[0][1], wh[G][1], ah[cur][0]) NC_MFMA_P(acc4, wh[G][CW], ah[cur][1])
        NC_MFMA_P(acc[0][0], wh[G][0], al[cur][0]) NC_MFMA_P(acc[0][1], wh[G][1], al[cur][0]) NC_MFMA_P(acc4, wh[G][CW], al[cur][1])
        NC_MFMA_P(acc[0][0], wl[G][0], ah[cur][0]) NC_MFMA_P(acc[0][1], wl[G][1], ah[cur][0]) NC_MFMA_P(acc4, wl[G][CW], ah[cur][1])
        __builtin_amdgcn_sched_barrier(0);
    }
    P3_T(3)
    // ---- phase B: tile CW + 2 (index 1), with the epilogue of phase A's accumulators in slices: K groups 0..5 take (SELU of an accumulator, then
    // its split + stores) in turn
    auto loadB = [&](int G, int slot) {
        const int off = (int)((c2_off_pack(G) >> sh) & 0xffffu);
        ah[slot][0] = lds_h8(A1H + abase[1] + off); al[slot][0] = lds_h8(A1H + abase[1] + off + T_A1PLANE);
    };
    loadB(0, 0);
    f32x4v sv[3];
#pragma unroll
    for (int G = 0; G < 9; G++) {
        const int cur = G % NB;
        if (G + DP < 9) loadB(G + DP, (G + DP) % NB);
        __builtin_amdgcn_sched_barrier(0);
        NC_MFMA_P(acc[1][0], wh[G][0], ah[cur][0]) NC_MFMA_P(acc[1][1], wh[G][1], ah[cur][0])
        NC_MFMA_P(acc[1][0], wh[G][0], al[cur][0]) NC_MFMA_P(acc[1][1], wh[G][1], al[cur][0])
        NC_MFMA_P(acc[1][0], wl[G][0], ah[cur][0]) NC_MFMA_P(acc[1][1], wl[G][1], ah[cur][0])
        if (G < 6) {
            const int a = G >> 1;                                     // accumulator 0, 1: tile CW's channel halves; 2: tile 4
            if ((G & 1) == 0) sv[a] = selu4_scaled(a == 0 ? acc[0][0] : a == 1 ? acc[0][1] : acc4, epi);
            else {
                const int oo = a < 2 ? obase[0] + a * 2 * T_PL2 * 8 : obase[2] + CW * 2 * T_PL2 * 8;
                split4_store(sv[a], A2H + oo, A2H + oo + T_A2PLANE);
            }
#pragma unroll
            for (int q = 0; q < 6; q++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, NC_PIPE_NV, 0);   // NC_PIPE_NV vector instructions
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    P3_T(1)
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        const int oo = obase[1] + tn * 2 * T_PL2 * 8;
        split4_store(selu4_scaled(acc[1][tn], epi), A2H + oo, A2H + oo + T_A2PLANE);
    }
}
template <int CW>
__device__ __forceinline__ void t_conv2_pair(const _Float16 *A1H, _Float16 *A2H, const h8 (&wh)[9][2], const h8 (&wl)[9][2],
                                             const float *__restrict__ b2s, const h_epi &epi, int lane, unsigned long long *trk = nullptr)
{
    c2_acc o;
    t_conv2_pair_mma<CW>(A1H, wh, wl, b2s, lane, o);
    P3_T(1)
    t_conv2_pair_epi<CW>(A2H, epi, lane, o);
}

// conv3, wave CW of two: output channels 32 CW .. 32 CW + 31 of both position tiles
struct c3_acc { f32x4v a[2][2]; };
template <int CW>
__device__ __forceinline__ void t_conv3_pair_mma(const _Float16 *A2H, const h8 (&w3h)[6][2], const h8 (&w3l)[6][2], const float *__restrict__ b3s, const int (&c3slot)[2], int lane,
                                                 c3_acc &o)
{
    const int g = lane >> 4;
    int abase[2];
#pragma unroll
    for (int tm = 0; tm < 2; tm++) {
        abase[tm] = (g * T_PL2 + c3slot[tm]) * 8;
        asm volatile("" : "+v"(abase[tm]));
    }
    f32x4v acc[2][2];
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
        acc[0][tn] = *reinterpret_cast<const f32x4v *>(b3s + (2 * CW + tn) * 16 + 4 * g);
        acc[1][tn] = acc[0][tn];
    }
#ifndef NC_P3_C3_DEPTH
#define NC_P3_C3_DEPTH 1
#endif
    constexpr int DP = NC_P3_C3_DEPTH, NB = DP + 1;
    h8 ah[NB][2], al[NB][2];
    auto load3 = [&](int G, int slot) {
        const int off = ((G / 3) * T_R2 + (G % 3)) * 8;
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
            ah[slot][tm] = lds_h8(A2H + abase[tm] + off);
            al[slot][tm] = lds_h8(A2H + abase[tm] + off + T_A2PLANE);
        }
    };
#pragma unroll
    for (int G = 0; G < DP; G++) load3(G, G % NB);
#pragma unroll
    for (int G = 0; G < 6; G++) {
        const int cur = G % NB;
        if (G + DP < 6) load3(G + DP, (G + DP) % NB);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 2; tm++) { NC_MFMA(acc[tm][0], w3h[G][0], ah[cur][tm]) NC_MFMA(acc[tm][1], w3h[G][1], ah[cur][tm]) }
#pragma unroll
        for (int tm = 0; tm < 2; tm++) { NC_MFMA(acc[tm][0], w3h[G][0], al[cur][tm]) NC_MFMA(acc[tm][1], w3h[G][1], al[cur][tm]) }
#pragma unroll
        for (int tm = 0; tm < 2; tm++) { NC_MFMA(acc[tm][0], w3l[G][0], ah[cur][tm]) NC_MFMA(acc[tm][1], w3l[G][1], ah[cur][tm]) }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++) o.a[tm][tn] = acc[tm][tn];
}
template <int CW>
__device__ __forceinline__ void t_conv3_pair_epi(const h_epi &epi, float *__restrict__ out_site, const int (&c3out)[2], int lane, const c3_acc &o)
{
    const int g = lane >> 4;
#pragma unroll
    for (int tm = 0; tm < 2; tm++)
        if (c3out[tm] >= 0) {
#pragma unroll
            for (int tn = 0; tn < 2; tn++)
                *reinterpret_cast<f32x4v *>(out_site + c3out[tm] * 64 + (2 * CW + tn) * 16 + 4 * g) = selu4_scaled(o.a[tm][tn], epi);
        }
}
template <int CW>
__device__ __forceinline__ void t_conv3_pair(const _Float16 *A2H, const h8 (&w3h)[6][2], const h8 (&w3l)[6][2], const float *__restrict__ b3s,
                                             const h_epi &epi, float *__restrict__ out_site, const int (&c3slot)[2], const int (&c3out)[2], int lane,
                                             unsigned long long *trk = nullptr)
{
    c3_acc o;
    t_conv3_pair_mma<CW>(A2H, w3h, w3l, b3s, c3slot, lane, o);
    P3_T(1)
    t_conv3_pair_epi<CW>(epi, out_site, c3out, lane, o);
}

// Three roles, one barrier per site (the default trunk since round 5).  Step s:
//   waves 4-7: conv1 of site s (X[s & 1] -> A1[s & 1]); waves 4, 5 three tiles (72 MFMAs), waves 6, 7 three tiles + their part of tile 12 (86 / 82);
//   waves 0, 1: conv2 of site s - 1 (A1[(s - 1) & 1] -> A2[(s - 1) & 1]), all 32 channels of two tiles + one channel half of tile 4 (135 MFMAs);
//   waves 2, 3: staging of site s + 1 into X[(s + 1) & 1] (two pixels per thread, loads issued a step earlier), then conv3 of site s - 2
//               (A2[s & 1] -> HBM; 32 of the 64 channels each, 72 MFMAs).
// Every wave's weight fragments stay in its registers (conv1 96, conv2 144, conv3 96 VGPRs).  Waves w and w + 4 share a SIMD (tools/ubench/simdmap.hip):
// 207 / 207 / 158 / 154 MFMAs per SIMD and site.  What sets the step is the longest single-wave chain {requests, MFMAs, epilogue}, not a pipe: the roles
// are cut so that the chains are even (per-wave phase times: NC_TRACE_P3 + tools/exp_trunk.py), and the MFMA-heaviest role sits on the OLDEST waves,
// whose instructions the issue arbiter prefers (a younger wave's MFMAs starve behind an older wave's vector burst, not the other way round).
template <bool X16>
__global__ __launch_bounds__(512) void k5_trunk_p3(const float *__restrict__ x, const uint8_t *__restrict__ wp, float *__restrict__ a3,
                                                   int64_t n_sites, const double *__restrict__ scale, int scale_mode, int64_t site0, float x_limit,
                                                   uint8_t *__restrict__ range_sites)
{
    __shared__ __attribute__((aligned(16))) _Float16 X[2][2 * T_XPLANE];
    __shared__ __attribute__((aligned(16))) _Float16 A1[2][2 * T_A1PLANE];
    __shared__ __attribute__((aligned(16))) _Float16 A2[2][2 * T_A2PLANE];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#ifdef NC_TRACE_BLOCKS
    if (threadIdx.x == 0 && blockIdx.x < 128) (&nc_trace_buf[0][0][0])[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef NC_TRACE_BLOCKS
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 3) (&nc_trace_buf[0][0][0])[504 + (threadIdx.x >> 6)] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | (1ull << 40);
#endif
    const uint4 *w1f = reinterpret_cast<const uint4 *>(wp), *w2h = w1f + T_NW1 * 64, *w2l = w2h + T_NW2 * 64, *w3h = w2l + T_NW2 * 64,
                *w3l = w3h + T_NW3 * 64;
    __shared__ __attribute__((aligned(16))) float BIAS[48 + 32 + 64];
    const float *bg = reinterpret_cast<const float *>(w3l + T_NW3 * 64);
    const float *b1s = BIAS, *b2s = BIAS + 48, *b3s = BIAS + 80;
    const int *c3tab = reinterpret_cast<const int *>(bg + 48 + 32 + 68);
    const float inv_s = bg[48 + 32 + 64];
    if (threadIdx.x < 48 + 32 + 64) BIAS[threadIdx.x] = bg[threadIdx.x];
    const h_epi epi = {inv_s * 1.44269504088896341f, inv_s * SELU_L, 60000.0f / (inv_s * SELU_L)};
    const int64_t n_k = (n_sites - blockIdx.x + gridDim.x - 1) / gridDim.x;        // sites of this workgroup (>= 1)
    for (int i = threadIdx.x; i < 4 * T_XS; i += 512) *reinterpret_cast<uint4 *>(&X[0][0] + i * 8) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // staging, by the two conv3 waves (the lightest role): thread t of the 128 owns pixels t and t + 128 (< 205).  prefetch() only issues the loads
    // of a site's tensor; commit() converts them a step later (scale, range guard, fp16 hi / lo) and writes the two operand planes of X
    auto stage_px = [&](int j) { const int t = (int)threadIdx.x - 128 + 128 * j; return t < 205 ? t : 204; };
    if (wv >= 4) {
        // ------------------------------------------------------------------ conv1 of site s: 13 tiles as 3 / 3 / 3 / 3 + tile 12 shared by waves 0, 1
        h8 w1[T_NW1];
#pragma unroll
        for (int q = 0; q < T_NW1; q++) w1[q] = as_h8(w1f[q * 64 + lane]);
        __syncthreads();                                                           // P0
        for (int64_t s = 0; s < n_k + 2; s++) {
#ifdef NC_TRACE_P3
            unsigned long long *trk = (blockIdx.x == 3 && lane == 0 && s >= 8 && s < 16) ? &nc_trace_buf[wv][s - 8][0] : nullptr;
#else
            unsigned long long *trk = nullptr;
#endif
            P3_T(0)
            P3_T(1)
            if (s < n_k) {
                const int buf = (int)(s & 1);
                // waves 4, 5 (beside the conv2 waves: 135 MFMAs) three tiles, waves 6, 7 (beside the conv3 waves: 72) three tiles + their part of tile 12
                if (wv == 6) t_conv1<4, 4>(X[buf], A1[buf], w1, b1s, epi, 0, lane, 12, 4, trk, 2);     // tiles 0, 4, 8, the 5x5 channels of tile 12
                else if (wv == 7) t_conv1<4, 3>(X[buf], A1[buf], w1, b1s, epi, 1, lane, 12, 4, trk, 2);  // tiles 1, 5, 9, the 1x5 + 5x1 channels of tile 12
                else t_conv1<3>(X[buf], A1[buf], w1, b1s, epi, wv - 2, lane, -1, 4, trk, 2);           // tiles 2, 6, 10 / 3, 7, 11
            }
            P3_T(5)
            NC_SITE_SYNC();
            P3_T(6)
        }
    } else if (wv < 2) {
        // ------------------------------------------------------------------ conv2 of site s - 1
        h8 c2h[9][2], c2l[9][2];
#pragma unroll
        for (int q = 0; q < 9; q++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) { c2h[q][tn] = as_h8(w2h[(q * 2 + tn) * 64 + lane]); c2l[q][tn] = as_h8(w2l[(q * 2 + tn) * 64 + lane]); }
        __syncthreads();                                                           // P0
        for (int64_t s = 0; s < n_k + 2; s++) {
#ifdef NC_TRACE_P3
            unsigned long long *trk = (blockIdx.x == 3 && lane == 0 && s >= 8 && s < 16) ? &nc_trace_buf[wv][s - 8][0] : nullptr;
#else
            unsigned long long *trk = nullptr;
#endif
            P3_T(0)
            if (s >= 1 && s - 1 < n_k) {
                const int buf = (int)((s - 1) & 1);
                if (wv == 0) t_conv2_pair<0>(A1[buf], A2[buf], c2h, c2l, b2s, epi, lane, trk);
                else t_conv2_pair<1>(A1[buf], A2[buf], c2h, c2l, b2s, epi, lane, trk);
            }
            P3_T(5)
            NC_SITE_SYNC();
            P3_T(6)
        }
    } else {
        // ------------------------------------------------------------------ staging of site s + 1, conv3 of site s - 2
        const int cw = wv - 2;
        h8 c3h[6][2], c3l[6][2];
#pragma unroll
        for (int q = 0; q < 6; q++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) {
                c3h[q][tn] = as_h8(w3h[(q * 4 + 2 * cw + tn) * 64 + lane]);
                c3l[q][tn] = as_h8(w3l[(q * 4 + 2 * cw + tn) * 64 + lane]);
            }
        const int c3slot[2] = {c3tab[lane & 15], c3tab[16 + (lane & 15)]}, c3out[2] = {c3tab[32 + (lane & 15)], c3tab[48 + (lane & 15)]};
        const int st = (int)threadIdx.x - 128;                                    // 0..127
        float pre[2][5];
        uint32_t raw[2][3];
        double pre_sd = 1.0;
        int64_t pre_site = 0;
        auto prefetch = [&](int64_t site) {
            pre_site = site;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int px = stage_px(j);
                if constexpr (X16) {
                    const int16_t *xs = reinterpret_cast<const int16_t *>(x) + site * NC_SNP_TENSOR + px * 5;       // 2-byte aligned
                    typedef uint32_t __attribute__((aligned(2))) u32_a2;
                    raw[j][0] = *reinterpret_cast<const u32_a2 *>(xs);
                    raw[j][1] = *reinterpret_cast<const u32_a2 *>(xs + 2);
                    raw[j][2] = (uint32_t)(uint16_t)xs[4];
                } else {
                    const float *xs = x + site * NC_SNP_TENSOR + px * 5;
#pragma unroll
                    for (int u = 0; u < 5; u++) pre[j][u] = xs[u];
                }
            }
            if (scale) pre_sd = scale[site0 + site];
        };
        auto commit = [&](int buf) {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (st + 128 * j >= 205) continue;
                const int px = st + 128 * j, ph = px / 41, pw = px - ph * 41;
                const int xslot = ((ph + 2) * T_RX + pw + 2) * 8;
                float v5[5];
                if constexpr (X16) {
                    v5[0] = (float)(int16_t)(raw[j][0] & 0xffffu); v5[1] = (float)(int16_t)(raw[j][0] >> 16);
                    v5[2] = (float)(int16_t)(raw[j][1] & 0xffffu); v5[3] = (float)(int16_t)(raw[j][1] >> 16);
                    v5[4] = (float)(int16_t)raw[j][2];
                } else {
#pragma unroll
                    for (int u = 0; u < 5; u++) v5[u] = pre[j][u];
                }
                const double md = (scale && ph > 0) ? pre_sd : 1.0;               // snpCaller.py:93-96
                const float mf = (float)md;
                if (scale_mode == 0) {
#pragma unroll
                    for (int u = 0; u < 4; u++) v5[u] *= mf;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++) v5[u] = (float)((double)v5[u] * md);
                }
                const float amax = fmaxf(fmaxf(fmaxf(fabsf(v5[0]), fabsf(v5[1])), fmaxf(fabsf(v5[2]), fabsf(v5[3]))), fabsf(v5[4]));
                if (range_sites && !(amax <= x_limit)) range_sites[site0 + pre_site] = 1;
                _Float16 hi[5], lo[5];
#pragma unroll
                for (int u = 0; u < 5; u++) {
                    float v = v5[u];
                    v = fminf(fmaxf(v, -65504.0f), 65504.0f);
                    hi[u] = (_Float16)v;
                    lo[u] = (_Float16)(v - (float)hi[u]);
                }
                const h8 sa = {hi[0], hi[1], hi[2], hi[3], hi[4], lo[0], lo[1], lo[2]};
                const h8 sb = {lo[3], lo[4], hi[0], hi[1], hi[2], hi[3], hi[4], (_Float16)0.0f};
                *reinterpret_cast<h8 *>(&X[buf][xslot]) = sa;
                *reinterpret_cast<h8 *>(&X[buf][xslot + T_XPLANE]) = sb;
            }
        };
        int64_t site = blockIdx.x;                                                 // the site conv3 works on next
        prefetch(blockIdx.x);
        commit(0);
        if (n_k > 1) prefetch((int64_t)blockIdx.x + gridDim.x);
        __syncthreads();                                                           // P0
        for (int64_t s = 0; s < n_k + 2; s++) {
#ifdef NC_TRACE_P3
            unsigned long long *trk = (blockIdx.x == 3 && lane == 0 && s >= 8 && s < 16) ? &nc_trace_buf[wv][s - 8][0] : nullptr;
#else
            unsigned long long *trk = nullptr;
#endif
            P3_T(0)
            // the other X buffer's last reader was conv1 of site s - 1 (a barrier ago); site s + 1's loads were issued a step ago
            if (s + 1 < n_k) commit((int)((s + 1) & 1));
            if (s + 2 < n_k) prefetch((int64_t)blockIdx.x + (s + 2) * gridDim.x);
            P3_T(2)
            if (s >= 2) {
                const int buf = (int)(s & 1);
                float *out_site = a3 + site * (27 * 64);
                if (cw == 0) t_conv3_pair<0>(A2[buf], c3h, c3l, b3s, epi, out_site, c3slot, c3out, lane, trk);
                else t_conv3_pair<1>(A2[buf], c3h, c3l, b3s, epi, out_site, c3slot, c3out, lane, trk);
                site += gridDim.x;
            }
            P3_T(5)
            NC_SITE_SYNC();
            P3_T(6)
        }
    }
}
// =====================================================================================================================
// k5_trunk_lin: the three-stage trunk with conv1 BY LINEARITY (round 6; VERDICT r5 #1b / #1c).
//
// The int16 site tensor is exact in fp16 (|x| <= 2048) and conv1 is linear, so the coverage scale s (snpCaller.py:93-96) does not have
// to be multiplied into the operand (which makes it a 22-bit number: hi + lo planes, three products): with
//     pre-activation = b + s * SUM(W x_scaled) + SUM(W u_unscaled) = s * [ b / s + SUM(W x) + SUM(W (u / s)) ]
// the scaled entries (rows 1-4, channels 0-3) enter the MFMA as the INTEGERS they are -- two products (x Wh + x Wl), no lo plane -- the few
// unscaled ones (channel 4 = the reference-base marker, and row 0 = the one-hot reference row) enter as u * rho, rho = 1 / s split into fp16
// hi / lo, and s itself folds into the epilogue's constants (the accumulators already carry a scale S).  Operand layout: a "record" per
// (image row r, column w) = the 5 pixels (r, w-2 .. w+2) x 6 values [x0 x1 x2 x3 hi(u4 rho) lo(u4 rho)] = 30 of 32 K slots = ONE K group per
// 5-tap kernel row (the pixel form of k5_trunk_p3 spends 7 K groups x 2 planes on the 25 taps); row 0 has its own records (hi and lo plane of
// u rho, same K layout, same weight fragments).  Kernel rows that fall on the `same` padding of the 5-row image for every position of a tile are
// not executed at all.  Per site: conv1 199 MFMAs (312 before) and 83 operand reads (182): 613 MFMAs (726), 239 ds_read_b128 (338).
// LDS: X buffer = P1 [4 slots][176 records] + P0 hi / lo [4][48] each (slot-planar: the 16 positions of a tile read 16 consecutive 16-byte slots).
#ifndef NC_LIN_PIPE
#define NC_LIN_PIPE 0                 // bit 0: conv2 in two phases with its first epilogue inside the second MFMA loop; bit 1: conv3; bit 2: conv1
#endif
constexpr int L_NR1 = 176, L_NR0 = 48;                                       // records per slot plane (multiples of 16): 4 x 41 = 164 + zero / dump records; 41 + ...
constexpr int L_P0H = 4 * L_NR1 * 8, L_P0L = L_P0H + 4 * L_NR0 * 8, L_XBUF = L_P0L + 4 * L_NR0 * 8;   // halves
constexpr int L_Z1 = 164, L_Z0 = 41, L_DUMP1 = 170, L_DUMP0 = 44;           // all-zero records (read by lanes whose kernel row is off the image); write-only dump records
constexpr int T_NW1L = 17;                                                   // conv1 fragments: 5x5 hi / lo per kernel row, 1x5 hi / lo, 5x1 groups A, B (hi / lo), C (hi)
enum { LW_5H = 0, LW_5L = 5, LW_1H = 10, LW_1L = 11, LW_2HA = 12, LW_2LA = 13, LW_2HB = 14, LW_2LB = 15, LW_2HC = 16 };
constexpr int L_PACKED_BYTES = 2 * T_FRAG * T_NW1L;
constexpr int L_DXO[5] = {0, -2, -1, 1, 2};                                  // pixel i of a record is column w + L_DXO[i] (the centre first: slot 0 serves the 5x1 kernel)
// which tiles of 16 positions a conv1 wave computes (position p = 16 t + lane % 16 = 41 h + w)
// `part` = which of a tile's three accumulators the wave computes (1: the 1x5 kernel's 16 channels, 2: the 5x1 kernel's, 4: the 5x5 kernel's): the 13th
// tile (positions 192..204: 10 MFMAs but a full 12 activations per lane) is split by kernel over three waves, 36 / 40 / 40 / 40 activations per lane
// instead of 48 / 36 / 36 / 36 -- the conv1 waves' epilogues are the pole of the step (profiles/r06_trunk_phases.md)
struct c1l_role { int nt; int tile[5]; int part[5]; };
#ifndef NC_LIN_ROLES
#define NC_LIN_ROLES {{3, {0, 4, 8, 0, 0}, {7, 7, 7, 0, 0}}, {4, {1, 5, 9, 12, 0}, {7, 7, 7, 1, 0}}, {4, {2, 6, 10, 12, 0}, {7, 7, 7, 2, 0}}, {4, {3, 7, 11, 12, 0}, {7, 7, 7, 4, 0}}}
#endif
constexpr c1l_role C1L_ROLES[4] = NC_LIN_ROLES;
constexpr int c1l_hlo(int t) { return (16 * t) / 41; }
constexpr int c1l_hhi(int t) { return (16 * t + 15 > 204 ? 204 : 16 * t + 15) / 41; }
constexpr bool c1l_v1(int h, int dy) { return h + dy - 2 >= 1 && h + dy - 2 <= 4; }
constexpr bool c1l_v0(int h, int dy) { return h + dy - 2 == 0; }
constexpr bool c1l_need1(int t, int dy) { return c1l_v1(c1l_hlo(t), dy) || c1l_v1(c1l_hhi(t), dy); }
constexpr bool c1l_all1(int t, int dy) { return c1l_v1(c1l_hlo(t), dy) && c1l_v1(c1l_hhi(t), dy); }
constexpr bool c1l_need0(int t, int dy) { return c1l_v0(c1l_hlo(t), dy) || c1l_v0(c1l_hhi(t), dy); }
constexpr bool c1l_all0(int t, int dy) { return c1l_v0(c1l_hlo(t), dy) && c1l_v0(c1l_hhi(t), dy); }
constexpr bool c1l_low(int t) { return c1l_hlo(t) <= 2; }                    // some position of the tile has image row 0 under a 5x1 tap (groups B, C)
constexpr int c1l_n0(int role, int dy) { int n = 0; for (int tm = 0; tm < C1L_ROLES[role].nt; tm++) n += c1l_need0(C1L_ROLES[role].tile[tm], dy) ? 1 : 0; return n; }
constexpr int c1l_mfma_tile(int t)
{
    int n = 0;
    for (int dy = 0; dy < 5; dy++) n += (c1l_need1(t, dy) ? 2 : 0) + (c1l_need0(t, dy) ? 3 : 0) + (dy == 2 ? (c1l_need1(t, dy) ? 2 : 0) + (c1l_need0(t, dy) ? 3 : 0) : 0);
    return n + 2 + (c1l_low(t) ? 3 : 0);
}
constexpr bool c1l_roles_cover()                                             // every accumulator of every tile is exactly one wave's
{
    for (int t = 0; t < 13; t++) {
        int seen = 0;
        for (int r = 0; r < 4; r++)
            for (int tm = 0; tm < C1L_ROLES[r].nt; tm++)
                if (C1L_ROLES[r].tile[tm] == t) { if (seen & C1L_ROLES[r].part[tm]) return false; seen |= C1L_ROLES[r].part[tm]; }
        if (seen != 7) return false;
    }
    return true;
}
static_assert(c1l_roles_cover(), "NC_LIN_ROLES: the roles do not cover the 13 tiles x 3 kernels exactly once");
constexpr int c1l_mfma_site() { int n = 0; for (int t = 0; t < 13; t++) n += c1l_mfma_tile(t); return n; }
constexpr int L_MFMA_PER_SITE = c1l_mfma_site() + 10 * 27 + 8 * 18;

// the MFMAs of tiles [T0, T0 + NTT) of a conv1 role; payload(step) is called once per step (7 steps) inside the step's scheduling region: the two-phase
// form passes the epilogue of the role's EARLIER tiles in slices, issued between this loop's MFMAs (NC_LIN_PIPE bit 2)
template <int ROLE, int T0, int NTT, typename P>
__device__ __forceinline__ void c1l_mma(const _Float16 *XB, const h8 (&wl)[T_NW1L], const float *__restrict__ b1s, float rho, int lane, f32x4v (&acc1)[NTT], f32x4v (&acc2)[NTT],
                                        f32x4v (&acc3)[NTT], int (&obase)[NTT], P payload)
{
    constexpr c1l_role R = C1L_ROLES[ROLE];
    constexpr int NT = NTT;
    auto tile_of = [](int tm) constexpr { return C1L_ROLES[ROLE].tile[T0 + tm]; };
    auto part_of = [](int tm) constexpr { return C1L_ROLES[ROLE].part[T0 + tm]; };
    auto k55 = [&](int tm) constexpr { return (part_of(tm) & 4) != 0; };            // the 5x5 kernel's accumulator of this tile is this wave's
    auto k15 = [&](int tm) constexpr { return (part_of(tm) & 1) != 0; };
    auto k51 = [&](int tm) constexpr { return (part_of(tm) & 2) != 0; };
    auto use = [&](int tm, int s) constexpr { return k55(tm) || (s == 2 && k15(tm)); };   // the kernel row's operand is needed
    const int g = lane >> 4, c16 = lane & 15;
    int pr8[NT], w8[NT], hrow[NT];
#pragma unroll
    for (int tm = 0; tm < NT; tm++) {
        const int t = tile_of(tm);
        const int p = 16 * t + c16, pr = p < 205 ? p : 204;
        const int h = c1l_hlo(t) == c1l_hhi(t) ? c1l_hlo(t) : (pr >= 41 * c1l_hhi(t) ? c1l_hhi(t) : c1l_hlo(t));
        const int w = pr - 41 * h;
        hrow[tm] = h;
        pr8[tm] = (pr - 123) * 8;                                      // halves; record of kernel row dy = pr - 123 + 41 dy  (= (h + dy - 3) * 41 + w)
        w8[tm] = w * 8;
        // positions 205..207 (tile 12) go to the three unused slots at the end of row 4
        obase[tm] = ((g >> 1) * T_PL1 + (p < 205 ? h * T_R1 + w : 4 * T_R1 + 41 + (p - 205))) * 8 + (g & 1) * 4;
        asm volatile("" : "+v"(pr8[tm]), "+v"(w8[tm]), "+v"(obase[tm]), "+v"(hrow[tm]));     // (keeps the address arithmetic inside the site loop: no spills)
    }
    (void)R;
    const int g1 = g * L_NR1 * 8, g0 = L_P0H + g * L_NR0 * 8;          // this lane group's slot plane
    auto a1 = [&](int tm, int dy) -> int {                             // operand of the 5-tap kernel row dy, rows 1..4
        const int t = tile_of(tm);
        int a = g1 + pr8[tm] + 41 * 8 * dy;
        if (!c1l_all1(t, dy)) { const int r = hrow[tm] + dy - 2; a = (r >= 1 && r <= 4) ? a : g1 + L_Z1 * 8; }
        return a;
    };
    auto a0 = [&](int tm, int dy) -> int {                             // the same of image row 0 (hi plane; the lo plane is L_P0L - L_P0H further)
        const int t = tile_of(tm);
        int a = g0 + w8[tm];
        if (!c1l_all0(t, dy)) a = (hrow[tm] + dy - 2 == 0) ? a : g0 + L_Z0 * 8;
        return a;
    };
    // 5x1 kernel: lane group g reads slot 0 (the centre pixel) of the record of ITS kernel row.  A: rows dy = g of P1.  B: g = 0: dy = 4 of P1;
    // g = 1..3: dy = g - 1 of the row-0 hi plane.  C: g = 0..2: dy = g of the row-0 lo plane.
    auto aA = [&](int tm) -> int { const int r = hrow[tm] + g - 2; return (r >= 1 && r <= 4) ? pr8[tm] + 41 * 8 * g : L_Z1 * 8; };
    auto aB = [&](int tm) -> int {
        const int h = hrow[tm];
        const int p1 = h <= 2 ? pr8[tm] + 41 * 8 * 4 : L_Z1 * 8, p0 = L_P0H + (h == 3 - g ? w8[tm] : L_Z0 * 8);
        return g == 0 ? p1 : p0;
    };
    auto aC = [&](int tm) -> int { return L_P0L + ((g < 3 && hrow[tm] == 2 - g) ? w8[tm] : L_Z0 * 8); };
    {
        const f32x4v x1 = *reinterpret_cast<const f32x4v *>(b1s + 4 * g) * rho, x2 = *reinterpret_cast<const f32x4v *>(b1s + 16 + 4 * g) * rho,
                     x3 = *reinterpret_cast<const f32x4v *>(b1s + 32 + 4 * g) * rho;
#pragma unroll
        for (int tm = 0; tm < NT; tm++) { acc1[tm] = x1; acc2[tm] = x2; acc3[tm] = x3; }
    }
    // steps 0..4: the 5-tap kernel rows dy (5x5 kernel; dy == 2 also feeds the 1x5 kernel); step 5: 5x1 group A; step 6: 5x1 groups B and C.
    // The operands of step s + DP are requested before the MFMAs of step s (register buffers).
#ifndef NC_LIN_DEPTH
#define NC_LIN_DEPTH 1
#endif
    constexpr int DP = NC_LIN_DEPTH, NB = DP + 1;                      // steps requested ahead of the one being multiplied / register buffers
    h8 x1[NB][NT], xh[NB][2], xl[NB][2], xc[NT];
    auto load = [&](int s, int slot) {
        if (s < 5) {
            int n0 = 0;
#pragma unroll
            for (int tm = 0; tm < NT; tm++) {
                const int t = tile_of(tm);
                if (!use(tm, s)) continue;
                if (c1l_need1(t, s)) x1[slot][tm] = lds_h8(XB + a1(tm, s));
                if (c1l_need0(t, s)) { const int a = a0(tm, s); xh[slot][n0] = lds_h8(XB + a); xl[slot][n0] = lds_h8(XB + a + (L_P0L - L_P0H)); n0++; }
            }
        } else if (s == 5) {
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (k51(tm)) x1[slot][tm] = lds_h8(XB + aA(tm));
        } else {
#pragma unroll
            for (int tm = 0; tm < NT; tm++)
                if (k51(tm) && c1l_low(tile_of(tm))) { x1[slot][tm] = lds_h8(XB + aB(tm)); xc[tm] = lds_h8(XB + aC(tm)); }
        }
    };
#pragma unroll
    for (int s = 0; s < DP; s++) load(s, s % NB);
#pragma unroll
    for (int s = 0; s < 7; s++) {
        const int cur = s % NB;
        if (s + DP < 7) load(s + DP, (s + DP) % NB);
        __builtin_amdgcn_sched_barrier(0);
        if (s < 5) {
            // independent accumulators interleaved; the three products of a row-0 operand pair are spread over the step
            // (n0: index of the tile's row-0 operand pair among those this wave loaded for the step)
            int n0 = 0;
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (k55(tm) && c1l_need1(tile_of(tm), s)) { NC_MFMA(acc3[tm], wl[LW_5H + s], x1[cur][tm]) }
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (use(tm, s) && c1l_need0(tile_of(tm), s)) { if (k55(tm)) { NC_MFMA(acc3[tm], wl[LW_5H + s], xh[cur][n0]) } if (s == 2 && k15(tm)) { NC_MFMA(acc1[tm], wl[LW_1H], xh[cur][n0]) } n0++; }
            if (s == 2) {
#pragma unroll
                for (int tm = 0; tm < NT; tm++) if (k15(tm) && c1l_need1(tile_of(tm), s)) { NC_MFMA(acc1[tm], wl[LW_1H], x1[cur][tm]) }
            }
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (k55(tm) && c1l_need1(tile_of(tm), s)) { NC_MFMA(acc3[tm], wl[LW_5L + s], x1[cur][tm]) }
            n0 = 0;
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (use(tm, s) && c1l_need0(tile_of(tm), s)) { if (k55(tm)) { NC_MFMA(acc3[tm], wl[LW_5H + s], xl[cur][n0]) } if (s == 2 && k15(tm)) { NC_MFMA(acc1[tm], wl[LW_1H], xl[cur][n0]) } n0++; }
            if (s == 2) {
#pragma unroll
                for (int tm = 0; tm < NT; tm++) if (k15(tm) && c1l_need1(tile_of(tm), s)) { NC_MFMA(acc1[tm], wl[LW_1L], x1[cur][tm]) }
            }
            n0 = 0;
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (use(tm, s) && c1l_need0(tile_of(tm), s)) { if (k55(tm)) { NC_MFMA(acc3[tm], wl[LW_5L + s], xh[cur][n0]) } if (s == 2 && k15(tm)) { NC_MFMA(acc1[tm], wl[LW_1L], xh[cur][n0]) } n0++; }
        } else if (s == 5) {
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (k51(tm)) { NC_MFMA(acc2[tm], wl[LW_2HA], x1[cur][tm]) }
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (k51(tm)) { NC_MFMA(acc2[tm], wl[LW_2LA], x1[cur][tm]) }
        } else {
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (k51(tm) && c1l_low(tile_of(tm))) { NC_MFMA(acc2[tm], wl[LW_2HB], x1[cur][tm]) }
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (k51(tm) && c1l_low(tile_of(tm))) { NC_MFMA(acc2[tm], wl[LW_2HC], xc[tm]) }
#pragma unroll
            for (int tm = 0; tm < NT; tm++) if (k51(tm) && c1l_low(tile_of(tm))) { NC_MFMA(acc2[tm], wl[LW_2LB], x1[cur][tm]) }
        }
        payload(s);
        __builtin_amdgcn_sched_barrier(0);
    }
}
// the epilogue of one accumulator of a conv1 tile: which = 0 the 1x5 kernel's 16 channels, 1 the 5x1 kernel's, 2 the 5x5 kernel's
__device__ __forceinline__ void c1l_epi_one(const f32x4v &acc, int which, int o, const h_epi &epi, _Float16 *A1H)
{
    split4_store(selu4_scaled(acc, epi), A1H + o + which * 2 * T_PL1 * 8, A1H + o + which * 2 * T_PL1 * 8 + T_A1PLANE);
}

template <int ROLE>
__device__ __forceinline__ void t_conv1_lin(const _Float16 *XB, _Float16 *A1H, const h8 (&wl)[T_NW1L], const float *__restrict__ b1s, const h_epi &epi, float rho, int lane,
                                            unsigned long long *trk = nullptr)
{
    constexpr int NT = C1L_ROLES[ROLE].nt;
    static_assert(c1l_n0(ROLE, 0) <= 2 && c1l_n0(ROLE, 1) <= 2 && c1l_n0(ROLE, 2) <= 2, "t_conv1_lin: at most two row-0 operand pairs per kernel row and wave");
#if NC_LIN_PIPE & 4
    // two phases: the MFMAs of the first two tiles; then the MFMAs of the others with the first two tiles' epilogue between them (one accumulator per step:
    // a wave's own vector instructions ride in the shadow of its own MFMAs); then the epilogue of the others
    constexpr int NA = 2, NB2 = NT - 2;
    static_assert(C1L_ROLES[ROLE].part[0] == 7 && C1L_ROLES[ROLE].part[1] == 7 && C1L_ROLES[ROLE].part[2] == 7 && (NT < 4 || C1L_ROLES[ROLE].part[3] == 7), "two-phase conv1: whole tiles only");
    f32x4v p1[NA], p2[NA], p3[NA], q1[NB2], q2[NB2], q3[NB2];
    int oa[NA], ob[NB2];
    c1l_mma<ROLE, 0, NA>(XB, wl, b1s, rho, lane, p1, p2, p3, oa, [](int) {});
    P3_T(3)
    c1l_mma<ROLE, 2, NB2>(XB, wl, b1s, rho, lane, q1, q2, q3, ob, [&](int s) {
        if (s < 6) {
            const int tm = s / 3, which = s % 3;
            c1l_epi_one(which == 0 ? p1[tm] : which == 1 ? p2[tm] : p3[tm], which, oa[tm], epi, A1H);
#pragma unroll
            for (int q = 0; q < 8; q++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, NC_PIPE_NV, 0);   // NC_PIPE_NV vector instructions
            }
        }
    });
    if (trk) trk[2] = __builtin_readcyclecounter();
#pragma unroll
    for (int tm = 0; tm < NB2; tm++) {
        c1l_epi_one(q1[tm], 0, ob[tm], epi, A1H);
        c1l_epi_one(q2[tm], 1, ob[tm], epi, A1H);
        c1l_epi_one(q3[tm], 2, ob[tm], epi, A1H);
    }
#else
    f32x4v acc1[NT], acc2[NT], acc3[NT];
    int obase[NT];
    c1l_mma<ROLE, 0, NT>(XB, wl, b1s, rho, lane, acc1, acc2, acc3, obase, [](int) {});
    if (trk) trk[2] = __builtin_readcyclecounter();
#pragma unroll
    for (int tm = 0; tm < NT; tm++) {
        if (C1L_ROLES[ROLE].part[tm] & 1) c1l_epi_one(acc1[tm], 0, obase[tm], epi, A1H);
        if (C1L_ROLES[ROLE].part[tm] & 2) c1l_epi_one(acc2[tm], 1, obase[tm], epi, A1H);
        if (C1L_ROLES[ROLE].part[tm] & 4) c1l_epi_one(acc3[tm], 2, obase[tm], epi, A1H);
    }
#endif
}

// fp16 hi / lo of a float as one packed dword (lo in the upper half)
__device__ __forceinline__ uint32_t split_pack(float v)
{
    v = fminf(fmaxf(v, -65504.0f), 65504.0f);
    const _Float16 h = (_Float16)v, l = (_Float16)(v - (float)h);
    return (uint32_t)__builtin_bit_cast(uint16_t, h) | ((uint32_t)__builtin_bit_cast(uint16_t, l) << 16);
}

__global__ __launch_bounds__(512) void k5_trunk_lin(const int16_t *__restrict__ x, const uint8_t *__restrict__ wp, const uint8_t *__restrict__ wlin, float *__restrict__ a3,
                                                    int64_t n_sites, const double *__restrict__ scale, int64_t site0, float x_limit, uint8_t *__restrict__ range_sites)
{
    __shared__ __attribute__((aligned(16))) _Float16 X[2][L_XBUF];
    __shared__ __attribute__((aligned(16))) _Float16 A1[2][2 * T_A1PLANE];
    __shared__ __attribute__((aligned(16))) _Float16 A2[2][2 * T_A2PLANE];
    __shared__ __attribute__((aligned(16))) float BIAS[48 + 32 + 64];
    // role of a hardware wave (roles 0, 1: conv2; 2, 3: staging + conv3; 4..7: conv1 roles 0..3).  Waves w and w + 4 share a SIMD and the lower-numbered one
    // is the older, which the issue arbiter prefers (MI355X_MICROARCH.md, two waves per SIMD): NC_LIN_PERM lists the role of hardware waves 0..7.
#ifndef NC_LIN_PERM
#define NC_LIN_PERM {2, 3, 4, 5, 6, 7, 0, 1}
#endif
    constexpr int LIN_PERM[8] = NC_LIN_PERM;
    const int lane = threadIdx.x & 63, wv = LIN_PERM[threadIdx.x >> 6];
#ifdef NC_LIN_PRIO                                                   // (experiment) static issue priority 1 for the roles whose bit is set
    if ((NC_LIN_PRIO >> wv) & 1) __builtin_amdgcn_s_setprio(1);
#endif
    const uint4 *w1f = reinterpret_cast<const uint4 *>(wp), *w2h = w1f + T_NW1 * 64, *w2l = w2h + T_NW2 * 64, *w3h = w2l + T_NW2 * 64, *w3l = w3h + T_NW3 * 64;
    const uint4 *wlf = reinterpret_cast<const uint4 *>(wlin);
    const float *bg = reinterpret_cast<const float *>(w3l + T_NW3 * 64);
    const float *b1s = BIAS, *b2s = BIAS + 48, *b3s = BIAS + 80;
    const int *c3tab = reinterpret_cast<const int *>(bg + 48 + 32 + 68);
    const float inv_s = bg[48 + 32 + 64];
    if (threadIdx.x < 48 + 32 + 64) BIAS[threadIdx.x] = bg[threadIdx.x];
    const h_epi epi = {inv_s * 1.44269504088896341f, inv_s * SELU_L, 60000.0f / (inv_s * SELU_L)};
    const int64_t n_k = (n_sites - blockIdx.x + gridDim.x - 1) / gridDim.x;        // sites of this workgroup (>= 1)
    for (int i = threadIdx.x; i < 2 * L_XBUF / 8; i += 512) *reinterpret_cast<uint4 *>(&X[0][0] + i * 8) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // coverage scale of a site (snpCaller.py:93-96): s as float (numpy's float32 product in both modes up to one rounding of the operand, which this
    // kernel does not perform at all), 1 without a scale array
    auto site_scale = [&](int64_t k) -> float { return scale ? (float)scale[site0 + blockIdx.x + k * gridDim.x] : 1.0f; };
    // NC_LIN_SKEW=1 (measured slower, profiles/r06_trunk_phases.md; off): the conv2 and conv3 waves run the EPILOGUE of their previous step's accumulators first and their MFMA loop second, the
    // conv1 waves their MFMA loop first and their epilogue second: on every SIMD one wave is in its vector phase while the other is in its matrix
    // phase (the two add up otherwise: per-wave phase times, profiles/r06_trunk_phases.md).  conv2's / conv3's results land a step later: one more step.
#ifndef NC_LIN_SKEW
#define NC_LIN_SKEW 0
#endif

    constexpr int SKEW = NC_LIN_SKEW;
    const int64_t n_steps = n_k + 2 + SKEW;
    if (wv >= 4) {
        // ------------------------------------------------------------------ conv1 of site s
        h8 wl[T_NW1L];
#pragma unroll
        for (int q = 0; q < T_NW1L; q++) wl[q] = as_h8(wlf[q * 64 + lane]);
        float s_next = site_scale(0);
        __syncthreads();                                                           // P0
        for (int64_t s = 0; s < n_steps; s++) {
#ifdef NC_TRACE_P3
            unsigned long long *trk = (blockIdx.x == 3 && lane == 0 && s >= 8 && s < 16) ? &nc_trace_buf[wv][s - 8][0] : nullptr;
#else
            unsigned long long *trk = nullptr;
#endif
            P3_T(0)
            P3_T(1)
            if (s < n_k) {
                const int buf = (int)(s & 1);
                const float sf = s_next, rho = 1.0f / sf;
                if (s + 1 < n_k) s_next = site_scale(s + 1);                       // (a scalar load: a step ahead of its use)
                const h_epi e1 = {epi.c1 * sf, epi.c2 * sf, epi.c3 * rho};         // the accumulators hold S / s x (pre-activation)
                if (wv == 4) t_conv1_lin<0>(X[buf], A1[buf], wl, b1s, e1, rho, lane, trk);
                else if (wv == 5) t_conv1_lin<1>(X[buf], A1[buf], wl, b1s, e1, rho, lane, trk);
                else if (wv == 6) t_conv1_lin<2>(X[buf], A1[buf], wl, b1s, e1, rho, lane, trk);
                else t_conv1_lin<3>(X[buf], A1[buf], wl, b1s, e1, rho, lane, trk);
            }
            P3_T(5)
            NC_SITE_SYNC();
            P3_T(6)
        }
    } else if (wv < 2) {
        // ------------------------------------------------------------------ conv2 of site s - 1
        h8 c2h[9][2], c2l[9][2];
#pragma unroll
        for (int q = 0; q < 9; q++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) { c2h[q][tn] = as_h8(w2h[(q * 2 + tn) * 64 + lane]); c2l[q][tn] = as_h8(w2l[(q * 2 + tn) * 64 + lane]); }
        c2_acc pend2;
        __syncthreads();                                                           // P0
        for (int64_t s = 0; s < n_steps; s++) {
#ifdef NC_TRACE_P3
            unsigned long long *trk = (blockIdx.x == 3 && lane == 0 && s >= 8 && s < 16) ? &nc_trace_buf[wv][s - 8][0] : nullptr;
#else
            unsigned long long *trk = nullptr;
#endif
            P3_T(0)
            if constexpr (SKEW) {
                const bool do_epi = s >= 2 && s - 2 < n_k, do_mma = s >= 1 && s - 1 < n_k;
                {
                if (do_epi) {                                                      // epilogue of site s - 2 (MFMA loop: a step ago)
                    const int buf = (int)(s & 1);
                    if (wv == 0) t_conv2_pair_epi<0>(A2[buf], epi, lane, pend2);
                    else t_conv2_pair_epi<1>(A2[buf], epi, lane, pend2);
                }
                P3_T(3)
                if (do_mma) {
                    const int buf = (int)((s - 1) & 1);
                    if (wv == 0) t_conv2_pair_mma<0>(A1[buf], c2h, c2l, b2s, lane, pend2);
                    else t_conv2_pair_mma<1>(A1[buf], c2h, c2l, b2s, lane, pend2);
                }
                P3_T(1)
                }
            } else if (s >= 1 && s - 1 < n_k) {
                const int buf = (int)((s - 1) & 1);
#if NC_LIN_PIPE & 1
                if (wv == 0) t_conv2_two_phase<0>(A1[buf], A2[buf], c2h, c2l, b2s, epi, lane, trk);
                else t_conv2_two_phase<1>(A1[buf], A2[buf], c2h, c2l, b2s, epi, lane, trk);
#else
                if (wv == 0) t_conv2_pair<0>(A1[buf], A2[buf], c2h, c2l, b2s, epi, lane, trk);
                else t_conv2_pair<1>(A1[buf], A2[buf], c2h, c2l, b2s, epi, lane, trk);
#endif
            }
            P3_T(5)
            NC_SITE_SYNC();
            P3_T(6)
        }
    } else {
        // ------------------------------------------------------------------ staging of site s + 1, conv3 of site s - 2
        const int cw = wv - 2;
        h8 c3h[6][2], c3l[6][2];
#pragma unroll
        for (int q = 0; q < 6; q++)
#pragma unroll
            for (int tn = 0; tn < 2; tn++) {
                c3h[q][tn] = as_h8(w3h[(q * 4 + 2 * cw + tn) * 64 + lane]);
                c3l[q][tn] = as_h8(w3l[(q * 4 + 2 * cw + tn) * 64 + lane]);
            }
        const int c3slot[2] = {c3tab[lane & 15], c3tab[16 + (lane & 15)]}, c3out[2] = {c3tab[32 + (lane & 15)], c3tab[48 + (lane & 15)]};
        const int st = (wv - 2) * 64 + lane;                                      // 0..127: pixels st and st + 128 (< 205) of the 5 x 41 image
        uint32_t raw[2][3];
        float pre_s = 1.0f;
        int64_t pre_site = 0;
        auto prefetch = [&](int64_t k) {                                           // issues the loads of the workgroup's k-th site; commit() converts them a step later
            const int64_t site = (int64_t)blockIdx.x + k * gridDim.x;
            pre_site = site;
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int t = st + 128 * j, px = t < 205 ? t : 204;
                const int16_t *xs = x + site * NC_SNP_TENSOR + px * 5;             // 2-byte aligned
                typedef uint32_t __attribute__((aligned(2))) u32_a2;
                raw[j][0] = *reinterpret_cast<const u32_a2 *>(xs);
                raw[j][1] = *reinterpret_cast<const u32_a2 *>(xs + 2);
                raw[j][2] = (uint32_t)(uint16_t)xs[4];
            }
            pre_s = site_scale(k);
        };
        auto commit = [&](int buf) {
            const float sf = pre_s, rho = 1.0f / sf;
            _Float16 *XB = &X[buf][0];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                if (st + 128 * j >= 205) continue;
                const int px = st + 128 * j, ph = px / 41, pw = px - ph * 41;
                const int v0 = (int16_t)(raw[j][0] & 0xffffu), v1 = (int16_t)(raw[j][0] >> 16), v2 = (int16_t)(raw[j][1] & 0xffffu), v3 = (int16_t)(raw[j][1] >> 16),
                          v4 = (int16_t)raw[j][2];
                const int m4 = max(max(abs(v0), abs(v1)), max(abs(v2), abs(v3)));
                // range guard: the scaled entries as the epilogue sees them (|x| s), the unscaled ones as they are; integers beyond fp16's exact range
                // (2048) cannot take this kernel at all.  Flagged sites are computed again by the exact fp32 trunk (nc_cnn_range_watch).
                // An unscaled entry travels as u * rho in fp16 hi + lo: it must stay inside fp16's range, and rho's lo half must not sink into the
                // subnormals (s <= 64; a scale that is not a positive finite number takes the exact trunk as well).
                const float un = ph > 0 ? fabsf((float)v4) : fmaxf((float)m4, fabsf((float)v4));
                const float amax = ph > 0 ? fmaxf((float)m4 * sf, un) : un;
                if (range_sites && (!(amax <= x_limit) || (ph > 0 && m4 > 2048) || !(un * rho <= 60000.0f) || !(sf > 0.0f && sf <= 64.0f)))
                    range_sites[site0 + pre_site] = 1;
                // the pixel's 6 K values as three dwords per plane
                uint32_t d[2][3];
                const uint32_t u4 = split_pack((float)v4 * rho);
                if (ph > 0) {
                    const h2 q01 = {(_Float16)(float)v0, (_Float16)(float)v1}, q23 = {(_Float16)(float)v2, (_Float16)(float)v3};      // exact: |v| <= 2048
                    d[0][0] = __builtin_bit_cast(uint32_t, q01); d[0][1] = __builtin_bit_cast(uint32_t, q23); d[0][2] = u4;              // [x0 x1 x2 x3 hi(u4 rho) lo(u4 rho)]
                } else {
                    const uint32_t u0 = split_pack((float)v0 * rho), u1 = split_pack((float)v1 * rho), u2 = split_pack((float)v2 * rho), u3 = split_pack((float)v3 * rho);
                    d[0][0] = (u0 & 0xffffu) | (u1 << 16); d[0][1] = (u2 & 0xffffu) | (u3 << 16); d[0][2] = u4 & 0xffffu;                // hi plane [h0 h1 h2 h3 h4 0]
                    d[1][0] = (u0 >> 16) | (u1 & 0xffff0000u); d[1][1] = (u2 >> 16) | (u3 & 0xffff0000u); d[1][2] = u4 >> 16;            // lo plane [l0 l1 l2 l3 l4 0]
                }
                // the pixel is element i of the records of columns w' = pw - L_DXO[i] of its row: K slots 6 i .. 6 i + 5 of that record
                const int nr = ph > 0 ? L_NR1 : L_NR0, rec0 = ph > 0 ? (ph - 1) * 41 : 0, dump = ph > 0 ? L_DUMP1 : L_DUMP0, pbase = ph > 0 ? 0 : L_P0H;
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const int wq = pw - L_DXO[i];
                    const int rec = (wq >= 0 && wq <= 40) ? rec0 + wq : dump;
                    constexpr int NPL = 2;
#pragma unroll
                    for (int pl = 0; pl < NPL; pl++) {
                        if (pl == 1 && ph > 0) continue;
                        _Float16 *rb = XB + pbase + pl * (L_P0L - L_P0H) + rec * 8;
                        const int k0 = 6 * i, sl = k0 >> 3, of = k0 & 7;            // first K slot: 16-byte slot sl, half `of` (0, 6, 4, 2, 0)
                        _Float16 *q0 = rb + sl * nr * 8 + of;
                        if (of == 0) { *reinterpret_cast<uint2 *>(q0) = make_uint2(d[pl][0], d[pl][1]); *reinterpret_cast<uint32_t *>(q0 + 4) = d[pl][2]; }
                        else if (of == 6) { *reinterpret_cast<uint32_t *>(q0) = d[pl][0]; *reinterpret_cast<uint2 *>(rb + (sl + 1) * nr * 8) = make_uint2(d[pl][1], d[pl][2]); }
                        else if (of == 4) { *reinterpret_cast<uint2 *>(q0) = make_uint2(d[pl][0], d[pl][1]); *reinterpret_cast<uint32_t *>(rb + (sl + 1) * nr * 8) = d[pl][2]; }
                        else { *reinterpret_cast<uint32_t *>(q0) = d[pl][0]; *reinterpret_cast<uint2 *>(q0 + 2) = make_uint2(d[pl][1], d[pl][2]); }
                    }
                }
            }
        };
        int64_t site = blockIdx.x;                                                 // the site conv3's epilogue stores next
        c3_acc pend3;
        prefetch(0);
        commit(0);
        if (n_k > 1) prefetch(1);
        __syncthreads();                                                           // P0
        for (int64_t s = 0; s < n_steps; s++) {
#ifdef NC_TRACE_P3
            unsigned long long *trk = (blockIdx.x == 3 && lane == 0 && s >= 8 && s < 16) ? &nc_trace_buf[wv][s - 8][0] : nullptr;
#else
            unsigned long long *trk = nullptr;
#endif
            P3_T(0)
            // the other X buffer's last reader was conv1 of site s - 1 (a barrier ago); site s + 1's loads were issued a step ago
            if (s + 1 < n_k) commit((int)((s + 1) & 1));
            if (s + 2 < n_k) prefetch(s + 2);
            P3_T(2)
            if constexpr (SKEW) {
                if (s >= 4) {                                                      // epilogue of site s - 4 (MFMA loop: a step ago)
                    float *out_site = a3 + site * (27 * 64);
                    if (cw == 0) t_conv3_pair_epi<0>(epi, out_site, c3out, lane, pend3);
                    else t_conv3_pair_epi<1>(epi, out_site, c3out, lane, pend3);
                    site += gridDim.x;
                }
                P3_T(3)
                if (s >= 3) {                                                      // conv3 of site s - 3: conv2's epilogue wrote A2[(s - 3) & 1] during step s - 1
                    const int buf = (int)((s - 3) & 1);
                    if (cw == 0) t_conv3_pair_mma<0>(A2[buf], c3h, c3l, b3s, c3slot, lane, pend3);
                    else t_conv3_pair_mma<1>(A2[buf], c3h, c3l, b3s, c3slot, lane, pend3);
                }
                P3_T(1)
            } else if (s >= 2) {
                const int buf = (int)(s & 1);
                float *out_site = a3 + site * (27 * 64);
                if (cw == 0) t_conv3_pair<0>(A2[buf], c3h, c3l, b3s, epi, out_site, c3slot, c3out, lane, trk);
                else t_conv3_pair<1>(A2[buf], c3h, c3l, b3s, epi, out_site, c3slot, c3out, lane, trk);
                site += gridDim.x;
            }
            P3_T(5)
            NC_SITE_SYNC();
            P3_T(6)
        }
        if constexpr (SKEW) {                                                      // the last site's epilogue
            float *out_site = a3 + site * (27 * 64);
            if (cw == 0) t_conv3_pair_epi<0>(epi, out_site, c3out, lane, pend3);
            else t_conv3_pair_epi<1>(epi, out_site, c3out, lane, pend3);
        }
    }
}

#undef NC_MFMA3
#undef NC_MFMA

// ---- fc1 on the same split-precision scheme: out[site][48] = selu(W^T a3 + b), K = 1728 = 54 groups of 32.
// A workgroup owns 64 sites (4 position... site tiles of 16) so that every weight fragment read from L2 feeds 4 MFMAs per
// product; the four waves split K and combine through LDS.  The fp32 activations are split into fp16 hi/lo on load (each
// lane reads the 8 K values of its site as two dwordx4: the four lane groups of a site cover one 128-byte line).
#ifndef NC_FC_TM
#define NC_FC_TM 4
#endif
constexpr int FC_K = 1728, FC_G = FC_K / 32, FC_TM = NC_FC_TM, FC_TN = 3;
constexpr int FC_PACKED_BYTES = 2 * FC_G * FC_TN * T_FRAG * 2 + 4 * 64;        // hi + lo fragments, then b*S[48], 1/S
__device__ __forceinline__ void split8(const float4 &a, const float4 &b, h8 &hi, h8 &lo)
{
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t uh[4], ul[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
        const float x0 = v[2 * p], x1 = v[2 * p + 1];                  // selu outputs clamped to fp16 range by k5_trunk_h3's conv3 epilogue
        const h2 h = __builtin_convertvector((f32x2v){x0, x1}, h2);
        uh[p] = __builtin_bit_cast(uint32_t, h);
        const f32x2v d = {sub_h_lo(x0, uh[p]), sub_h_hi(x1, uh[p])};
        ul[p] = __builtin_bit_cast(uint32_t, (h2)__builtin_convertvector(d, h2));
    }
    hi = as_h8(make_uint4(uh[0], uh[1], uh[2], uh[3]));
    lo = as_h8(make_uint4(ul[0], ul[1], ul[2], ul[3]));
}

__global__ __launch_bounds__(256) void k6_fc1_h3(const float *__restrict__ in, const uint8_t *__restrict__ wp, float *__restrict__ out, int64_t n)
{
    __shared__ f32x4v red[3][FC_TM][FC_TN][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const uint4 *wh = reinterpret_cast<const uint4 *>(wp), *wl = wh + FC_G * FC_TN * 64;
    const float *bs = reinterpret_cast<const float *>(wl + FC_G * FC_TN * 64);
    const float inv_s = bs[48];
    const int64_t tile0 = (int64_t)blockIdx.x * (FC_TM * 16);
    const float *ip[FC_TM];
#pragma unroll
    for (int tm = 0; tm < FC_TM; tm++) {
        int64_t s = tile0 + tm * 16 + c16;
        if (s >= n) s = n - 1;
        ip[tm] = in + s * FC_K + 8 * g;
    }
    f32x4v acc[FC_TM][FC_TN];
#pragma unroll
    for (int tn = 0; tn < FC_TN; tn++) {
        f32x4v b = {0.0f, 0.0f, 0.0f, 0.0f};
        if (wv == 0) b = *reinterpret_cast<const f32x4v *>(bs + tn * 16 + 4 * g);
#pragma unroll
        for (int tm = 0; tm < FC_TM; tm++) acc[tm][tn] = b;
    }
    const int j0 = (FC_G * wv) / 4, j1 = (FC_G * (wv + 1)) / 4;
#pragma unroll 2
    for (int G = j0; G < j1; G++) {
        float4 a0[FC_TM], a1[FC_TM];
#pragma unroll
        for (int tm = 0; tm < FC_TM; tm++) {
            a0[tm] = *reinterpret_cast<const float4 *>(ip[tm] + 32 * G);
            a1[tm] = *reinterpret_cast<const float4 *>(ip[tm] + 32 * G + 4);
        }
        h8 bh[FC_TN], bl[FC_TN];
#pragma unroll
        for (int tn = 0; tn < FC_TN; tn++) {
            bh[tn] = as_h8(wh[(G * FC_TN + tn) * 64 + lane]);
            bl[tn] = as_h8(wl[(G * FC_TN + tn) * 64 + lane]);
        }
#pragma unroll
        for (int tm = 0; tm < FC_TM; tm++) {
            h8 xh, xl;
            split8(a0[tm], a1[tm], xh, xl);
#pragma unroll
            for (int tn = 0; tn < FC_TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[tn], xh, acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tn = 0; tn < FC_TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[tn], xl, acc[tm][tn], 0, 0, 0);
#pragma unroll
            for (int tn = 0; tn < FC_TN; tn++) acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[tn], xh, acc[tm][tn], 0, 0, 0);
        }
    }
    if (wv > 0) {
#pragma unroll
        for (int tm = 0; tm < FC_TM; tm++)
#pragma unroll
            for (int tn = 0; tn < FC_TN; tn++) red[wv - 1][tm][tn][lane] = acc[tm][tn];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int tm = 0; tm < FC_TM; tm++) {
            const int64_t s = tile0 + tm * 16 + c16;                   // D[channel 4g + r][site c16]
#pragma unroll
            for (int tn = 0; tn < FC_TN; tn++) {
                const f32x4v v = (acc[tm][tn] + red[0][tm][tn][lane] + red[1][tm][tn][lane] + red[2][tm][tn][lane]) * inv_s;
                if (s < n) *reinterpret_cast<f32x4v *>(out + s * 48 + tn * 16 + 4 * g) = (f32x4v){selu(v[0]), selu(v[1]), selu(v[2]), selu(v[3])};
            }
        }
    }
}

__device__ __forceinline__ void dense_small(const float *in, int n_in, const float *k, const float *b, int n_out, float *out, bool act)
{
    for (int o = 0; o < n_out; o++) {
        float acc = b[o];
        for (int i = 0; i < n_in; i++) acc = fmaf(in[i], k[i * n_out + o], acc);
        out[o] = act ? selu_acc(acc) : acc;
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

// ---- conv2 / conv3 of the indel models on the split-precision scheme of the SNP trunk: every fp32 product is
// hi*hi + hi*lo + lo*hi of fp16 halves on v_mfma_f32_16x16x32_f16 with fp32 accumulation (weights pre-scaled by a power of two
// and split on the host; activations arrive split from the producing layer).  Implicit GEMM, M = CO (weights = A operand,
// fragments in LDS), N = 16 output positions per wave iteration, K = 6 taps x CI walked as chunks of 8 input channels
// (one 16-byte load from each plane per lane), four chunks per MFMA, the tail padded with zeros (CI = 24: 18 chunks -> 5 MFMAs).
// A lane's accumulator registers are 4 consecutive output channels of ONE position: 8-byte (fp16 planes) or 16-byte (fp32)
// stores.  in: planes [site][HI][WI][CI]; out: planes [site][HO][WO][CO] (lo plane = hi plane + npos*CO), or fp32.
template <int CI, int CO>
struct H3Layer {
    static constexpr int NCH = 6 * CI / 8, NG = (NCH + 3) / 4, TN = CO / 16;
    static constexpr size_t FRAG_HALVES = (size_t)NG * TN * 64 * 8;
    static constexpr size_t BYTES = 2 * FRAG_HALVES * 2 + 4 * (CO + 4);       // hi + lo fragments, bias * S [CO], 1 / S, pad
};

template <int HI, int WI, int CI, int CO, bool OUT_F32>
__global__ __launch_bounds__(256) void k8_conv23_h3(const _Float16 *__restrict__ in_hi, const _Float16 *__restrict__ in_lo,
                                                    const uint8_t *__restrict__ wp, void *__restrict__ out, int64_t npos_in, int64_t npos)
{
    typedef H3Layer<CI, CO> LY;
    constexpr int HO = HI - 1, WO = (WI - 3) / 2 + 1, NCH = LY::NCH, NG = LY::NG, TN = LY::TN, C8 = CI / 8;
    static_assert(CI % 8 == 0 && CO % 16 == 0, "k8_conv23_h3: shape");
    __shared__ uint4 wfh[NG * TN * 64], wfl[NG * TN * 64];
    const uint4 *gh = reinterpret_cast<const uint4 *>(wp), *gl = gh + NG * TN * 64;
    const float *bs = reinterpret_cast<const float *>(gl + NG * TN * 64);
    for (int i = threadIdx.x; i < NG * TN * 64; i += 256) { wfh[i] = gh[i]; wfl[i] = gl[i]; }
    __syncthreads();
    const float inv_s = bs[CO];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    // element offset of this lane's chunk of every MFMA step relative to the top-left input pixel of an output position
    int toff[NG];
    bool tval[NG];
#pragma unroll
    for (int G = 0; G < NG; G++) {
        const int chunk = 4 * G + g, tap = chunk / C8, c8 = chunk - tap * C8;
        tval[G] = chunk < NCH;
        toff[G] = tval[G] ? ((tap / 3) * WI + (tap % 3)) * CI + 8 * c8 : 0;
    }
    f32x4v bias[TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) bias[tn] = *reinterpret_cast<const f32x4v *>(bs + 16 * tn + 4 * g);
    const int64_t ntiles = (npos + 15) / 16;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wv; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
        int64_t m = tile * 16 + c16;
        if (m >= npos) m = npos - 1;
        const int64_t site = m / (HO * WO);
        const int r = (int)(m - site * (HO * WO));
        const int y = r / WO, xq = r - y * WO;
        const int64_t base = ((site * HI + y) * WI + 2 * xq) * CI;
        f32x4v acc[TN];
#pragma unroll
        for (int tn = 0; tn < TN; tn++) acc[tn] = bias[tn];
        h8 xh[NG], xl[NG];
#pragma unroll
        for (int G = 0; G < NG; G++) {
            const uint4 z = make_uint4(0, 0, 0, 0);
            xh[G] = as_h8(tval[G] ? *reinterpret_cast<const uint4 *>(in_hi + base + toff[G]) : z);
            xl[G] = as_h8(tval[G] ? *reinterpret_cast<const uint4 *>(in_lo + base + toff[G]) : z);
        }
#pragma unroll
        for (int G = 0; G < NG; G++) {
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                const h8 wh = as_h8(wfh[(G * TN + tn) * 64 + lane]), wl = as_h8(wfl[(G * TN + tn) * 64 + lane]);
                acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[G], acc[tn], 0, 0, 0);
                acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[G], acc[tn], 0, 0, 0);
                acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[G], acc[tn], 0, 0, 0);
            }
        }
        const int64_t pos = tile * 16 + c16;                          // D[channel 4g + r][position c16]
        if (pos < npos) {
#pragma unroll
            for (int tn = 0; tn < TN; tn++) {
                f32x4v v;
#pragma unroll
                for (int q = 0; q < 4; q++) v[q] = selu(acc[tn][q] * inv_s);
                if constexpr (OUT_F32) {
                    *reinterpret_cast<f32x4v *>(reinterpret_cast<float *>(out) + pos * CO + 16 * tn + 4 * g) = v;
                } else {
                    _Float16 *hp = reinterpret_cast<_Float16 *>(out) + pos * CO + 16 * tn + 4 * g, *lp = hp + npos * CO;
                    _Float16 hi[4], lo[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float c = fminf(fmaxf(v[q], -65504.0f), 65504.0f);
                        hi[q] = (_Float16)c;
                        lo[q] = (_Float16)(c - (float)hi[q]);
                    }
                    *reinterpret_cast<uint2 *>(hp) = *reinterpret_cast<const uint2 *>(hi);
                    *reinterpret_cast<uint2 *>(lp) = *reinterpret_cast<const uint2 *>(lo);
                }
            }
        }
    }
    (void)npos_in;
}

// ---- conv1 of the indel models (CI = 2, three `same` kernels 1x5 / 5x1 / 5x5 with 8 filters each) on the matrix pipe, same
// split-precision scheme.  A tap needs 6 K slots: [w_hi w_hi | w_hi w_hi | w_lo w_lo] x [x_hi(c0) x_hi(c1) | x_lo(c0) x_lo(c1) |
// x_hi(c0) x_hi(c1)], so one kernel row of five taps is ONE v_mfma_f32_16x16x32_f16 (30 of its 32 K slots).  With a pixel kept
// in LDS as three dwords [H, L, H] (H = its two channels' hi halves, L = the lo halves), the 32 K values of an output position are
// 16 CONSECUTIVE dwords starting at its leftmost tap: lane (position, quarter g) reads dwords 3 x + 4 g .. + 3.  The A operand of
// kernel row dy carries the 5x5 filters in rows 0-7 and, in rows 8-15, the 5x1 filters on its centre tap; a sixth MFMA on the
// centre row carries the 1x5 filters: 6 MFMAs per 16 positions for all 24 channels.  One workgroup per site, rows top to
// bottom through a five-row ring in LDS (every input row is staged once).  The kernel that does this, k9_conv12_h3, runs conv2
// on the rows as they appear.
constexpr int C1H_ROWPX = 134;                                  // pixels -2 .. 131 of a row (zero padded)
constexpr size_t C1H_FRAG7 = 6 * 64 * 16 + 4 * 40;            // 6 A fragments, then S*bias[32] (acc35 rows 0-15, acc1 rows 0-15), 1/S
constexpr size_t C1H_BYTES = C1H_FRAG7 + 64 * 16;             // + the 1x5 fragment with its filters in rows 8-15 (k10_indel_trunk_h3)

// conv1 + conv2 of one site per workgroup iteration: conv1's output rows never leave the chip -- they go, as hi / lo fp16
// planes, into a two-row LDS ring from which conv2 (2x3 taps, stride 2 in x: output row r needs conv1 rows r and r+1) reads
// its B operands (one ds_read_b128 of 8 channels per plane and MFMA step).  conv2's weight fragments sit in LDS as in
// k8_conv23_h3.  Per image row y: conv1 row y -> ring; conv2 row y-1 (63 positions: one 16-position tile per wave).
// out: conv2's activations as fp16 planes [site][H-1][63][32] (lo plane = hi plane + npos2 * 32).
template <int H>
__global__ __launch_bounds__(256, 3) void k9_conv12_h3(const float *__restrict__ x, const uint8_t *__restrict__ wp1, const uint8_t *__restrict__ wp2,
                                                    void *__restrict__ out, int64_t n_sites, int64_t npos2)
{
    constexpr int W = 128, HO = H - 1, WO = 63, C1C = 24, CO = 32;
    typedef H3Layer<24, 32> LY;
    constexpr int NG = LY::NG, TN = LY::TN;
    __shared__ uint32_t X3[5 * C1H_ROWPX * 3 + 4];
    __shared__ __attribute__((aligned(16))) _Float16 A1H[2 * W * C1C], A1L[2 * W * C1C];      // conv1 rows (slot = row & 1): hi / lo planes
    __shared__ uint4 wfh[NG * TN * 64], wfl[NG * TN * 64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const uint4 *wf = reinterpret_cast<const uint4 *>(wp1);
    const float *bs = reinterpret_cast<const float *>(wf + 6 * 64);
    h8 A[6];
#pragma unroll
    for (int f = 0; f < 6; f++) A[f] = as_h8(wf[f * 64 + lane]);
    const f32x4v b35 = *reinterpret_cast<const f32x4v *>(bs + 4 * g), b1 = *reinterpret_cast<const f32x4v *>(bs + 16 + 4 * g);
    const float inv_s1 = bs[32];
    const uint4 *gh = reinterpret_cast<const uint4 *>(wp2), *gl = gh + NG * TN * 64;
    const float *bs2 = reinterpret_cast<const float *>(gl + NG * TN * 64);
    for (int i = threadIdx.x; i < NG * TN * 64; i += 256) { wfh[i] = gh[i]; wfl[i] = gl[i]; }
    const float inv_s2 = bs2[CO];
    f32x4v bias2[TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) bias2[tn] = *reinterpret_cast<const f32x4v *>(bs2 + 16 * tn + 4 * g);
    // conv2: LDS offset (halves) of this lane's chunk of every MFMA step relative to (ring row 0, input pixel 2 xq)
    int toff[NG], trow[NG];
    bool tval[NG];
#pragma unroll
    for (int G = 0; G < NG; G++) {
        const int chunk = 4 * G + g, tap = chunk / 3, c8 = chunk - tap * 3;
        tval[G] = chunk < LY::NCH;
        trow[G] = tval[G] ? tap / 3 : 0;
        toff[G] = tval[G] ? (tap % 3) * C1C + 8 * c8 : 0;
    }
    _Float16 *hp0 = reinterpret_cast<_Float16 *>(out), *lp0 = hp0 + npos2 * CO;
    auto stage_row = [&](int64_t site, int iy) {
        const int slot = (iy + 5) % 5;
        for (int p = threadIdx.x; p < C1H_ROWPX; p += 256) {
            const int px = p - 2;
            uint32_t Hh = 0, Ll = 0;
            if (iy >= 0 && iy < H && px >= 0 && px < W) {
                const float2 v = *reinterpret_cast<const float2 *>(x + ((site * H + iy) * W + px) * 2);
                const float v0 = fminf(fmaxf(v.x, -65504.0f), 65504.0f), v1 = fminf(fmaxf(v.y, -65504.0f), 65504.0f);
                const h2 hh = __builtin_convertvector((f32x2v){v0, v1}, h2);
                Hh = __builtin_bit_cast(uint32_t, hh);
                const f32x2v d = {v0 - (float)hh[0], v1 - (float)hh[1]};
                Ll = __builtin_bit_cast(uint32_t, (h2)__builtin_convertvector(d, h2));
            }
            uint32_t *q = X3 + (slot * C1H_ROWPX + p) * 3;
            q[0] = Hh; q[1] = Ll; q[2] = Hh;
        }
    };
    for (int64_t site = blockIdx.x; site < n_sites; site += gridDim.x) {
        __syncthreads();
        for (int iy = -2; iy <= 2; iy++) stage_row(site, iy);
        for (int y = 0; y < H; y++) {
            __syncthreads();                                          // input rows y-2 .. y+2 staged; conv2 of row y-2 done with ring slot y & 1
            // ---- conv1, row y -> ring slot y & 1
#pragma unroll
            for (int tt = 0; tt < 2; tt++) {
                const int xx = 16 * (2 * wv + tt) + c16;
                f32x4v acc35 = b35, acc1 = b1;
#pragma unroll
                for (int dy = 0; dy < 5; dy++) {
                    const uint32_t *q = X3 + (((y + dy - 2 + 5) % 5) * C1H_ROWPX) * 3 + 3 * xx + 4 * g;
                    const h8 B = as_h8(make_uint4(q[0], q[1], q[2], q[3]));
                    acc35 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[dy], B, acc35, 0, 0, 0);
                    if (dy == 2) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[5], B, acc1, 0, 0, 0);
                }
                const int o = ((y & 1) * W + xx) * C1C;
                auto put = [&](const f32x4v &acc, int ch0) {
                    _Float16 hi[4], lo[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float v = fminf(fmaxf(selu(acc[r] * inv_s1), -65504.0f), 65504.0f);
                        hi[r] = (_Float16)v;
                        lo[r] = (_Float16)(v - (float)hi[r]);
                    }
                    *reinterpret_cast<uint2 *>(A1H + o + ch0) = *reinterpret_cast<const uint2 *>(hi);
                    *reinterpret_cast<uint2 *>(A1L + o + ch0) = *reinterpret_cast<const uint2 *>(lo);
                };
                put(acc35, g < 2 ? 16 + 4 * g : 8 + 4 * (g - 2));
                if (g < 2) put(acc1, 4 * g);
            }
            __syncthreads();                                          // conv1 row y is in the ring; input row y-2 is free
            if (y + 1 < H) stage_row(site, y + 3);
            // ---- conv2, output row y-1 (conv1 rows y-1 and y), positions 16 wv .. 16 wv + 15
            if (y >= 1) {
                const int r2 = y - 1, xq = 16 * wv + c16, xc = xq < WO ? xq : WO - 1;
                f32x4v acc[TN];
#pragma unroll
                for (int tn = 0; tn < TN; tn++) acc[tn] = bias2[tn];
                h8 xh[NG], xl[NG];
#pragma unroll
                for (int G = 0; G < NG; G++) {
                    const int o = ((((r2 + trow[G]) & 1) * W) + 2 * xc) * C1C + toff[G];
                    const uint4 z = make_uint4(0, 0, 0, 0);
                    xh[G] = as_h8(tval[G] ? *reinterpret_cast<const uint4 *>(A1H + o) : z);
                    xl[G] = as_h8(tval[G] ? *reinterpret_cast<const uint4 *>(A1L + o) : z);
                }
#pragma unroll
                for (int G = 0; G < NG; G++) {
#pragma unroll
                    for (int tn = 0; tn < TN; tn++) {
                        const h8 wh = as_h8(wfh[(G * TN + tn) * 64 + lane]), wl = as_h8(wfl[(G * TN + tn) * 64 + lane]);
                        acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[G], acc[tn], 0, 0, 0);
                        acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[G], acc[tn], 0, 0, 0);
                        acc[tn] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[G], acc[tn], 0, 0, 0);
                    }
                }
                if (xq < WO) {
                    const int64_t pos = (site * HO + r2) * WO + xq;
#pragma unroll
                    for (int tn = 0; tn < TN; tn++) {
                        _Float16 hi[4], lo[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const float c = fminf(fmaxf(selu(acc[tn][q] * inv_s2), -65504.0f), 65504.0f);
                            hi[q] = (_Float16)c;
                            lo[q] = (_Float16)(c - (float)hi[q]);
                        }
                        *reinterpret_cast<uint2 *>(hp0 + pos * CO + 16 * tn + 4 * g) = *reinterpret_cast<const uint2 *>(hi);
                        *reinterpret_cast<uint2 *>(lp0 + pos * CO + 16 * tn + 4 * g) = *reinterpret_cast<const uint2 *>(lo);
                    }
                }
            }
        }
    }
}

// ---- the whole conv trunk of the indel models in one kernel, one workgroup per CU, twelve waves with fixed roles (three per SIMD:
// one of each).  The rows of a site (and of the sites after it: the workgroup's sites form one stream of P = H + 3 rows each --
// two zero rows, the H image rows, one zero row) move through three LDS rings, two rows per step; at step T
//     wave 11    writes input rows 2T+4, 2T+5 into the X3 ring (requested from HBM eight rows earlier),
//     waves 0-3  conv1 of rows 2T, 2T+1 (two 16-pixel tiles each: 24 MFMAs; the six input rows are read once for both) -> ring R1,
//     waves 4-7  conv2 of the row pairs starting at conv1 rows 2T-4 and 2T-3 (one 16-position tile, both channel tiles: 60 MFMAs)
//                -> ring R2,
//     waves 8-10 conv3 of the row pairs starting at conv2 rows 2T-8 and 2T-7 (one channel tile each, both position tiles: 72 MFMAs)
//                -> HBM (fp32, fc1's input),
// every role reading only what earlier steps wrote: ONE barrier per step, and the weight fragments of a wave's role stay in its
// registers for the whole launch.  Two rows per step give every wave two independent accumulation chains per tile (a step of one
// row was latency-bound: 1.04 ms per 13 k sites against 0.6 for this form), and P even keeps the pairs aligned with the sites: a
// pair is either skipped or computed whole (conv1 and conv3 have an odd number of rows per site: one row in 16 / 14 is computed and dropped).
// k9_conv12_h3 / k8_conv23_h3 re-read the weights from LDS for every tile (60 KB per wave and row) and passed conv2's activations
// through HBM (226 KB per site).
constexpr int T_P1 = 24, T_P2 = 40;                            // pixel pitch (halves) of rings R1 / R2: 8 consecutive lanes of a b128 read hit 8 distinct bank groups
constexpr int K10_NS = 6;                                       // slots of R1 / R2 (rows live at a time: the two a role writes + the four its reader is behind)
constexpr size_t K10_LDS = 8 * C1H_ROWPX * 12 + 64 + 2 * (K10_NS * 128 * T_P1 * 2) + 2 * (K10_NS * 64 * T_P2 * 2);

#ifdef NC_K10_NOBAR
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif
// a loaded weight fragment the compiler may not re-load inside the step loop (it otherwise sinks the loads into the loop to reach an
// occupancy the kernel's LDS use rules out anyway)
__device__ __forceinline__ void pin(h8 &v) { asm volatile("" : "+v"(v)); }

// ablation switches (tools/exp_build.sh): a role that only keeps the barrier
#ifdef NC_K10_SKIP
constexpr bool K10_C1 = !(NC_K10_SKIP & 1), K10_C2 = !(NC_K10_SKIP & 2), K10_C3 = !(NC_K10_SKIP & 4);
#else
constexpr bool K10_C1 = true, K10_C2 = true, K10_C3 = true;
#endif
#ifdef NC_K10_NOMFMA
#define K10_MFMA(ACC, A_, B_) asm volatile("" ::"v"(A_), "v"(B_));
#else
#ifdef NC_K10_NOP
#define K10_MFMA(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_, B_, ACC, 0, 0, 0); asm volatile("s_nop %1" : "+v"(ACC) : "n"(NC_K10_NOP));
#else
#define K10_MFMA(ACC, A_, B_) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_, B_, ACC, 0, 0, 0);
#endif
#endif
template <int H>
__global__ __launch_bounds__(768) void k10_indel_trunk_h3(const float *__restrict__ x, const uint8_t *__restrict__ wp1, const uint8_t *__restrict__ wp2,
                                                          const uint8_t *__restrict__ wp3, float *__restrict__ a3, int64_t n_sites)
{
    constexpr int W = 128, P = H + 3, WO2 = 63, WO3 = 31, HO3 = H - 2, NS = K10_NS;
    static_assert(P % 2 == 0, "k10_indel_trunk_h3: the stream period must be even");
    typedef H3Layer<24, 32> L2;
    typedef H3Layer<32, 48> L3;
    static_assert(L2::NG == 5 && L2::TN == 2 && L3::NG == 6 && L3::TN == 3, "k10_indel_trunk_h3: shape");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *X3 = reinterpret_cast<uint32_t *>(smem);                                           // 8 slots x 134 pixels x [H L H]
    constexpr int R1S = W * T_P1, R2S = 64 * T_P2;                                                // halves per ring slot
    _Float16 *R1H = reinterpret_cast<_Float16 *>(smem + 8 * C1H_ROWPX * 12 + 64), *R1L = R1H + NS * R1S;
    _Float16 *R2H = R1L + NS * R1S, *R2L = R2H + NS * R2S;
    // role of a wave: wv 0-3 conv1, 4-7 conv2, 8-10 conv3, 11 stager.  NC_K10_ORDER (experiment) permutes which HARDWARE waves (age = issue priority;
    // waves w, w + 4, w + 8 share a SIMD) take which role, keeping every role's SIMDs
#ifndef NC_K10_ORDER
#define NC_K10_ORDER 0
#endif
    const int lane = threadIdx.x & 63, hwv = threadIdx.x >> 6, g = lane >> 4, c16 = lane & 15;
    const int wv = NC_K10_ORDER == 1 ? (hwv < 4 ? hwv + 4 : hwv < 8 ? hwv - 4 : hwv)                  // conv2 | conv1 | conv3 + stager
                 : NC_K10_ORDER == 2 ? (hwv < 4 ? hwv + 8 : hwv < 8 ? hwv - 4 : hwv - 4)              // conv3 + stager | conv1 | conv2
                 : NC_K10_ORDER == 3 ? (hwv < 4 ? hwv + 8 : hwv < 8 ? hwv : hwv - 8)                  // conv3 + stager | conv2 | conv1
                 : NC_K10_ORDER == 4 ? (hwv < 4 ? hwv + 4 : hwv < 8 ? hwv + 4 : hwv - 8)              // conv2 | conv3 + stager | conv1
                 : hwv;
    const int nloc = (int)((n_sites - blockIdx.x + gridDim.x - 1) / gridDim.x);                   // this workgroup's sites: blockIdx.x + k gridDim.x
    constexpr int RD = 8;                                             // the input stager's look-ahead (rows); T is a multiple of RD / 2
    const int T = ((nloc * P + 8) / 2 + RD / 2 - 1) / (RD / 2) * (RD / 2);
    if (wv < 4) {
        // ---------------- conv1.  acc: rows 0-7 the 5x5 filters, rows 8-15 the 5x1 filters (centre column of A[dy]); accp: the 1x5
        // filters of BOTH tiles (A[5] has them in rows 0-7, A[6] in rows 8-15: lanes g < 2 end up with tile 0's pixel, g >= 2 with tile 1's)
        const uint4 *wf = reinterpret_cast<const uint4 *>(wp1);
        const float *bs = reinterpret_cast<const float *>(wf + 6 * 64);
        h8 A[7];
#pragma unroll
        for (int f = 0; f < 6; f++) A[f] = as_h8(wf[f * 64 + lane]);
        A[6] = as_h8(reinterpret_cast<const uint4 *>(wp1 + C1H_FRAG7)[lane]);
#pragma unroll
        for (int f = 0; f < 7; f++) pin(A[f]);
        const f32x4v b35 = *reinterpret_cast<const f32x4v *>(bs + 4 * g), b1 = *reinterpret_cast<const f32x4v *>(bs + 16 + 4 * (g & 1));
        const float inv_s1 = bs[32];
        const h_epi e1 = {inv_s1 * 1.44269504088896341f, inv_s1 * SELU_L, 60000.0f / (inv_s1 * SELU_L)};
        const int xx0 = 32 * wv + c16;
        const int o35 = xx0 * T_P1 + (g < 2 ? 16 + 4 * g : 8 + 4 * (g - 2)), op = (xx0 + 16 * (g >> 1)) * T_P1 + 4 * (g & 1);
        const uint32_t *xb = X3 + 3 * xx0 + 4 * g;
        for (int t = 0; t < T; t++) {
            const int u0 = 2 * t;
            if (K10_C1 && u0 % P != 0 && u0 / P < nloc) {                      // conv1 of stream rows u0, u0 + 1 (input rows u0 - 2 .. u0 + 3)
                f32x4v acc[2][2] = {{b35, b35}, {b35, b35}}, accp[2] = {b1, b1};
#pragma unroll
                for (int ir = 0; ir < 6; ir++) {
                    const uint32_t *q0 = xb + (((u0 + ir - 2) & 7) * C1H_ROWPX) * 3, *q1 = q0 + 48;
                    const h8 B0 = as_h8(make_uint4(q0[0], q0[1], q0[2], q0[3])), B1 = as_h8(make_uint4(q1[0], q1[1], q1[2], q1[3]));
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const int dy = ir - r;
                        if (dy < 0 || dy > 4) continue;
                        K10_MFMA(acc[r][0], A[dy], B0)
                        K10_MFMA(acc[r][1], A[dy], B1)
                        if (dy == 2) {
                            K10_MFMA(accp[r], A[5], B0)
                            K10_MFMA(accp[r], A[6], B1)
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int o = ((u0 + r) % NS) * R1S;
                    split4_store(selu4_scaled(acc[r][0], e1), R1H + o + o35, R1L + o + o35);
                    split4_store(selu4_scaled(acc[r][1], e1), R1H + o + o35 + 16 * T_P1, R1L + o + o35 + 16 * T_P1);
                    split4_store(selu4_scaled(accp[r], e1), R1H + o + op, R1L + o + op);
                }
            }
            lds_barrier();
        }
    } else if (wv < 8) {
        // ---------------- conv2, position tile wv - 4
        const uint4 *gh = reinterpret_cast<const uint4 *>(wp2), *gl = gh + L2::NG * L2::TN * 64;
        const float *bs2 = reinterpret_cast<const float *>(gl + L2::NG * L2::TN * 64);
        h8 wh[L2::NG][L2::TN], wl[L2::NG][L2::TN];
#pragma unroll
        for (int G = 0; G < L2::NG; G++)
#pragma unroll
            for (int tn = 0; tn < L2::TN; tn++) {
                wh[G][tn] = as_h8(gh[(G * L2::TN + tn) * 64 + lane]);
                wl[G][tn] = as_h8(gl[(G * L2::TN + tn) * 64 + lane]);
                pin(wh[G][tn]);
                pin(wl[G][tn]);
            }
        const float inv_s2 = bs2[32];
        const h_epi e2 = {inv_s2 * 1.44269504088896341f, inv_s2 * SELU_L, 60000.0f / (inv_s2 * SELU_L)};
        f32x4v bias2[L2::TN];
#pragma unroll
        for (int tn = 0; tn < L2::TN; tn++) bias2[tn] = *reinterpret_cast<const f32x4v *>(bs2 + 16 * tn + 4 * g);
        const int xq = 16 * (wv - 4) + c16, xc = xq < WO2 ? xq : WO2 - 1;
        int toff[L2::NG], trow[L2::NG];
        bool tval[L2::NG];
#pragma unroll
        for (int G = 0; G < L2::NG; G++) {
            const int chunk = 4 * G + g, tap = chunk / 3, c8 = chunk - tap * 3;
            tval[G] = chunk < L2::NCH;
            trow[G] = tval[G] ? tap / 3 : 0;
            toff[G] = tval[G] ? (2 * xc + tap % 3) * T_P1 + 8 * c8 : 0;
        }
        const int oq = xq * T_P2 + 4 * g;
        for (int t = 0; t < T; t++) {
            const int v0 = 2 * t - 4, s2 = v0 % P;                             // an even lag: the 14 (H - 1) conv2 rows of a site are whole pairs
            if (K10_C2 && v0 >= 0 && s2 >= 2 && s2 <= P - 4 && v0 / P < nloc) {  // conv2 rows of the conv1 stream rows (v0, v0+1) and (v0+1, v0+2)
                const _Float16 *rp[3];                                         // this lane's pixel in the three conv1 rows
#pragma unroll
                for (int r = 0; r < 3; r++) rp[r] = R1H + ((v0 + r) % NS) * R1S;
                f32x4v acc[2][L2::TN];
#pragma unroll
                for (int r = 0; r < 2; r++)
#pragma unroll
                    for (int tn = 0; tn < L2::TN; tn++) acc[r][tn] = bias2[tn];
#pragma unroll
                for (int G = 0; G < L2::NG; G++) {
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        // K slots past the last tap (G = 4, g >= 2) read the row's first pixel: their weights are zero
                        const _Float16 *q = (trow[G] ? rp[r + 1] : rp[r]) + toff[G];
                        const h8 xh = as_h8(*reinterpret_cast<const uint4 *>(q)), xl = as_h8(*reinterpret_cast<const uint4 *>(q + NS * R1S));
#pragma unroll
                        for (int tn = 0; tn < L2::TN; tn++) {
                            K10_MFMA(acc[r][tn], wh[G][tn], xh)
                            K10_MFMA(acc[r][tn], wh[G][tn], xl)
                            K10_MFMA(acc[r][tn], wl[G][tn], xh)
                        }
                    }
                }
                if (xq < WO2) {
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const int o = ((v0 + r) % NS) * R2S + oq;
#pragma unroll
                        for (int tn = 0; tn < L2::TN; tn++) split4_store(selu4_scaled(acc[r][tn], e2), R2H + o + 16 * tn, R2L + o + 16 * tn);
                    }
                }
            }
            lds_barrier();
        }
    } else if (wv < 11) {
        // ---------------- conv3, channel tile tn = wv - 8
        const int tn = wv - 8;
        const uint4 *gh = reinterpret_cast<const uint4 *>(wp3), *gl = gh + L3::NG * L3::TN * 64;
        const float *bs3 = reinterpret_cast<const float *>(gl + L3::NG * L3::TN * 64);
        h8 wh[L3::NG], wl[L3::NG];
#pragma unroll
        for (int G = 0; G < L3::NG; G++) {
            wh[G] = as_h8(gh[(G * L3::TN + tn) * 64 + lane]);
            wl[G] = as_h8(gl[(G * L3::TN + tn) * 64 + lane]);
            pin(wh[G]);
            pin(wl[G]);
        }
        const float inv_s3 = bs3[48];
        const h_epi e3 = {inv_s3 * 1.44269504088896341f, inv_s3 * SELU_L, 3.0e38f};
        const f32x4v bias3 = *reinterpret_cast<const f32x4v *>(bs3 + 16 * tn + 4 * g);
        const int xqb = 16 + c16, xcb = xqb < WO3 ? xqb : WO3 - 1;
        const int oa = 2 * c16 * T_P2 + 8 * g, ob = 2 * xcb * T_P2 + 8 * g;
        for (int t = 0; t < T; t++) {
            const int w0 = 2 * t - 8, s3 = w0 % P;
            if (K10_C3 && w0 >= 0 && s3 != 0 && s3 < P - 2 && w0 / P < nloc) {  // conv3 rows of the conv2 stream rows (w0, w0+1) and (w0+1, w0+2)
                const int64_t site = blockIdx.x + (int64_t)(w0 / P) * gridDim.x;
                const _Float16 *pa[3], *pb[3];                                  // this lane's two pixels in the three conv2 rows
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    pa[r] = R2H + ((w0 + r) % NS) * R2S + oa;
                    pb[r] = R2H + ((w0 + r) % NS) * R2S + ob;
                }
                f32x4v acc[2][2] = {{bias3, bias3}, {bias3, bias3}};
#pragma unroll
                for (int G = 0; G < L3::NG; G++) {                             // tap G = (row G / 3, column G % 3), channels 8 g .. 8 g + 7
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const _Float16 *qa = pa[r + G / 3] + (G % 3) * T_P2, *qb = pb[r + G / 3] + (G % 3) * T_P2;
                        const h8 xh0 = as_h8(*reinterpret_cast<const uint4 *>(qa)), xl0 = as_h8(*reinterpret_cast<const uint4 *>(qa + NS * R2S));
                        const h8 xh1 = as_h8(*reinterpret_cast<const uint4 *>(qb)), xl1 = as_h8(*reinterpret_cast<const uint4 *>(qb + NS * R2S));
                        K10_MFMA(acc[r][0], wh[G], xh0)
                        K10_MFMA(acc[r][1], wh[G], xh1)
                        K10_MFMA(acc[r][0], wh[G], xl0)
                        K10_MFMA(acc[r][1], wh[G], xl1)
                        K10_MFMA(acc[r][0], wl[G], xh0)
                        K10_MFMA(acc[r][1], wl[G], xh1)
                    }
                }
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    if (s3 + r > P - 4) continue;                              // the pair's second row is past the site's last conv3 row
                    float *o = a3 + ((site * HO3 + (s3 + r - 2)) * WO3) * 48 + 16 * tn + 4 * g;
                    *reinterpret_cast<f32x4v *>(o + c16 * 48) = selu4_scaled(acc[r][0], e3);
                    if (xqb < WO3) *reinterpret_cast<f32x4v *>(o + xqb * 48) = selu4_scaled(acc[r][1], e3);
                }
            }
            lds_barrier();
        }
    } else {
        // ---------------- input rows: stream row u = [zero, zero, row 0 .. row H-1, zero] of the workgroup's sites, then zeros.
        // A row is requested RD rows before it is written into X3: HBM latency (~1-2 us) is several steps long.
        float2 rgs[RD][3];
        // the loads are unconditional (clamped address, value masked when stored) and the loop below has no branch: a load under a
        // branch makes the compiler wait for vmcnt(0) at every step, i.e. for the rows it has just requested
        auto load_row = [&](int u, float2 *rg) {
            const int k = min(u / P, nloc - 1), iy = min(max(u % P - 2, 0), H - 1);
            const float *row = x + ((blockIdx.x + (int64_t)k * gridDim.x) * H + iy) * (W * 2);
#pragma unroll
            for (int j = 0; j < 3; j++) rg[j] = *reinterpret_cast<const float2 *>(row + 2 * min(max(lane + 64 * j - 2, 0), W - 1));
        };
        auto store_row = [&](int u, const float2 *rg) {
            const bool row_ok = u / P < nloc && u % P >= 2 && u % P < H + 2;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int p = min(lane + 64 * j, C1H_ROWPX - 1);                // lanes past the row rewrite its last pixel (a zero pad: same value)
                const bool ok = row_ok && p >= 2 && p < W + 2;
                const float v0 = ok ? fminf(fmaxf(rg[j].x, -65504.0f), 65504.0f) : 0.0f, v1 = ok ? fminf(fmaxf(rg[j].y, -65504.0f), 65504.0f) : 0.0f;
                const h2 hh = __builtin_convertvector((f32x2v){v0, v1}, h2);
                const uint32_t Hh = __builtin_bit_cast(uint32_t, hh);
                const f32x2v d = {sub_h_lo(v0, Hh), sub_h_hi(v1, Hh)};
                const uint32_t Ll = __builtin_bit_cast(uint32_t, (h2)__builtin_convertvector(d, h2));
                uint32_t *q = X3 + ((u & 7) * C1H_ROWPX + p) * 3;
                q[0] = Hh; q[1] = Ll; q[2] = Hh;
            }
        };
        for (int u = 0; u < 4; u++) { load_row(u, rgs[0]); store_row(u, rgs[0]); }
#pragma unroll
        for (int d = 0; d < RD; d++) load_row(4 + d, rgs[d]);
        for (int t0 = 0; t0 < T; t0 += RD / 2) {                             // rows 0-3 are first read at step 1, after the barrier of step 0
#pragma unroll
            for (int d = 0; d < RD; d += 2) {
                const int u = 2 * t0 + 4 + d;                                 // step t0 + d / 2 writes rows u, u + 1
                store_row(u, rgs[d]);
                store_row(u + 1, rgs[d + 1]);
                load_row(u + RD, rgs[d]);
                load_row(u + 1 + RD, rgs[d + 1]);
                lds_barrier();
            }
        }
    }
}
#undef K10_MFMA

// host: A fragments of conv1 (k9_conv12_h3) from the canonical conv1 weights (k11 [5][2][8], k12 [5][2][8], k13 [25][2][8] + biases)
inline void pack_conv1_h3(const float *w, uint8_t *dst)
{
    const float *k11 = w, *b11 = k11 + 5 * 2 * 8, *k12 = b11 + 8, *b12 = k12 + 5 * 2 * 8, *k13 = b12 + 8, *b13 = k13 + 25 * 2 * 8;
    float wmax = 0.0f;
    for (const float *q = w; q < b13 + 8; q++) wmax = std::fmax(wmax, std::fabs(*q));
    float S = 4096.0f;
    while (S > 1.0f && wmax * S > 16384.0f) S *= 0.5f;
    _Float16 *fr = reinterpret_cast<_Float16 *>(dst);
    float *bs = reinterpret_cast<float *>(dst + 6 * 64 * 16);
    for (int f = 0; f < 6; f++)
        for (int lane = 0; lane < 64; lane++)
            for (int j = 0; j < 8; j++) {
                const int g = lane >> 4, c = lane & 15, s = 8 * g + j, t = s / 6, r = s % 6, ci = r & 1;
                float wv = 0.0f;
                if (s < 30) {
                    if (f < 5) {
                        if (c < 8) wv = k13[((f * 5 + t) * 2 + ci) * 8 + c];
                        else if (t == 2) wv = k12[(f * 2 + ci) * 8 + (c - 8)];
                    } else if (c < 8) wv = k11[(t * 2 + ci) * 8 + c];
                }
                const float sv = wv * S;
                const _Float16 hh = (_Float16)sv, ll = (_Float16)(sv - (float)hh);
                fr[((size_t)f * 64 + lane) * 8 + j] = r < 4 ? hh : ll;
            }
    for (int c = 0; c < 8; c++) { bs[c] = b13[c] * S; bs[8 + c] = b12[c] * S; bs[16 + c] = b11[c] * S; bs[24 + c] = 0.0f; }
    bs[32] = 1.0f / S;
    _Float16 *f7 = reinterpret_cast<_Float16 *>(dst + C1H_FRAG7);            // fragment 5 moved down by eight rows
    for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 8; j++) {
            const int c = lane & 15;
            f7[(size_t)lane * 8 + j] = c >= 8 ? fr[((size_t)5 * 64 + (lane - 8)) * 8 + j] : (_Float16)0.0f;
        }
}

constexpr size_t INDEL_H3_BYTES = H3Layer<24, 32>::BYTES + H3Layer<32, 48>::BYTES + C1H_BYTES;

// host: fragments of one layer (canonical weights k[6][CI][CO], bias[CO]) into `dst`
template <int CI, int CO>
void pack_h3_layer(const float *k, const float *b, uint8_t *dst)
{
    typedef H3Layer<CI, CO> LY;
    float wmax = 0.0f;
    for (int i = 0; i < 6 * CI * CO; i++) wmax = std::fmax(wmax, std::fabs(k[i]));
    float S = 4096.0f;
    while (S > 1.0f && wmax * S > 16384.0f) S *= 0.5f;
    _Float16 *fh = reinterpret_cast<_Float16 *>(dst), *fl = fh + LY::FRAG_HALVES;
    float *bs = reinterpret_cast<float *>(fl + LY::FRAG_HALVES);
    for (int G = 0; G < LY::NG; G++)
        for (int tn = 0; tn < LY::TN; tn++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int g = lane >> 4, c = lane & 15, chunk = 4 * G + g;
                    float sv = 0.0f;
                    if (chunk < LY::NCH) {
                        const int tap = chunk / (CI / 8), ci = 8 * (chunk % (CI / 8)) + j;
                        sv = k[((size_t)tap * CI + ci) * CO + tn * 16 + c] * S;
                    }
                    const _Float16 h = (_Float16)sv;
                    const size_t o = ((size_t)(G * LY::TN + tn) * 64 + lane) * 8 + j;
                    fh[o] = h;
                    fl[o] = (_Float16)(sv - (float)h);
                }
    for (int c = 0; c < CO; c++) bs[c] = b[c] * S;
    bs[CO] = 1.0f / S;
}

const size_t NPARAM[4] = {109370, 108308, 634420, 158185};

inline unsigned blocks_for(int64_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// conv trunk for `nb` sites -> fc1 activations [nb][F]; *f1_out / *tail receive the fc1 buffer and the tail weights
// which split-precision SNP trunk a forward call runs: k5_trunk_lin (conv1 by linearity; int16 tensors only) unless NC_TRUNK_LIN=0 or the
// two-stage kernel was asked for (NC_TRUNK_P3=0); float32 tensors take k5_trunk_p3 (their entries need not be integers)
static bool trunk_lin_selected(const nc_ctx *ctx)
{
    const char *el = getenv("NC_TRUNK_LIN"), *e3 = getenv("NC_TRUNK_P3");
    return ctx->x_i16 && !ctx->cnn_exact_fp32 && !(el && el[0] == '0') && !(e3 && e3[0] == '0');
}

template <int H, int W, int CI, int C1, int C2, int C3, int F, int P2, int P3, bool MFMA>
int run_trunk(nc_ctx *ctx, const float *w, const float *packed, const uint8_t *packed_h, int64_t site0, int64_t nb, const float *x_batch, const double *scale, int scale_mode,
              const float **f1_out, const float **tail, float x_limit = 0.0f)
{
    constexpr int H2 = H - 1, W2 = (W - 3) / 2 + 1, H3 = H2 - 1, W3 = (W2 - 3) / 2 + 1;
    constexpr int64_t n1 = (int64_t)H * W * 3 * C1, n2 = (int64_t)H2 * W2 * C2, n3 = (int64_t)H3 * W3 * C3;
    // indel models: exact fp32 MFMA (k7) or, by default, the split-precision kernels fed with fp16 hi/lo planes: one fused kernel
    // (k10_indel_trunk_h3; NC_INDEL_TRUNK_SPLIT=1 keeps round 2's pair k9_conv12_h3 + k8_conv23_h3 for comparison)
    const bool indel_h3 = !MFMA && !ctx->cnn_exact_fp32 && packed_h != nullptr;
    static const bool trunk_split = getenv("NC_INDEL_TRUNK_SPLIT") != nullptr;
    const bool indel_fused = indel_h3 && !trunk_split && CI == 2 && C1 == 8 && W == 128;
    if constexpr (!MFMA)
        if (!indel_h3) NC_TRY(nc_ensure(ctx, ctx->cnn_a, (size_t)(nb * n1) * 4));
    if (!indel_fused) NC_TRY(nc_ensure(ctx, ctx->cnn_b, (size_t)(nb * n2) * 4));
    NC_TRY(nc_ensure(ctx, ctx->cnn_c, (size_t)(nb * (n3 + F)) * 4 + 64));
    float *a1 = (float *)ctx->cnn_a.p, *a2 = (float *)ctx->cnn_b.p, *a3 = (float *)ctx->cnn_c.p;
    float *f1 = a3 + ((nb * n3 + 3) & ~int64_t(3));
    const float *k2 = w + (5 + 5 + 25) * CI * C1 + 3 * C1, *b2 = k2 + 2 * 3 * 3 * C1 * C2;
    const float *k3 = b2 + C2, *b3 = k3 + 2 * 3 * C2 * C3;
    const float *kf = b3 + C3, *bf = kf + n3 * F;
    *tail = bf + F;
    *f1_out = f1;
    const int64_t np1 = nb * H * W, np2 = nb * H2 * W2, np3 = nb * H3 * W3;
    if constexpr (!MFMA) {
        if constexpr (CI == 2 && W % 4 == 0 && C1 == 8) {
            if (indel_fused) {
                constexpr size_t LDS = K10_LDS;
                bool &attr_set = ctx->k10_lds_set[H == 15 ? 0 : 1];          // per context (= per device): the attribute belongs to the device's copy of the function
                if (!attr_set) {
                    NC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&k10_indel_trunk_h3<H>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS));
                    attr_set = true;
                }
                hipLaunchKernelGGL((k10_indel_trunk_h3<H>), dim3((unsigned)(nb < 256 ? nb : 256)), dim3(768), LDS, ctx->stream, x_batch,
                                   packed_h + H3Layer<24, 32>::BYTES + H3Layer<32, 48>::BYTES, packed_h, packed_h + H3Layer<24, 32>::BYTES, a3, nb);
            } else if (indel_h3)                                  // conv1 + conv2 fused: conv1's activations stay in LDS
                hipLaunchKernelGGL((k9_conv12_h3<H>), dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, ctx->stream, x_batch,
                                   packed_h + H3Layer<24, 32>::BYTES + H3Layer<32, 48>::BYTES, packed_h, (void *)a2, nb, np2);
            else
                hipLaunchKernelGGL((k2_conv1_x4<H, W, C1, false>), dim3(blocks_for(np1 / 4)), dim3(256), 0, ctx->stream, x_batch, w, a1, np1);
        } else {
            hipLaunchKernelGGL((k2_conv1<H, W, CI, C1>), dim3(blocks_for(np1)), dim3(256), 0, ctx->stream, x_batch, w, a1, np1, scale,
                               scale_mode, site0);
        }
    }
    if constexpr (MFMA) {
        constexpr int TMF = 1;
        (void)np2;
        const unsigned nblk = (unsigned)(nb < 512 ? nb : 512);          // k4: 2 resident workgroups per CU, persistent over sites
#ifndef NC_EXP_NBLK
#define NC_EXP_NBLK 256
#endif
        const unsigned nblk5 = (unsigned)(nb < NC_EXP_NBLK ? nb : NC_EXP_NBLK);        // k5: one 512-thread workgroup per CU
        (void)np3; (void)a2; (void)k3; (void)b3;
        const bool tk = ctx->timing && ctx->n_kev + 2 <= 128;
        // timing mode: the start / stop events ride on the kernel's own dispatch packet (hipExtLaunchKernelGGL), so they
        // read the kernel's execution time and put no barrier packets between the launches of a batch
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (tk) {
            for (int e = 0; e < 2; e++)
                if (!ctx->kev[ctx->n_kev + e]) NC_HIP(ctx, hipEventCreate(&ctx->kev[ctx->n_kev + e]));
            ev0 = ctx->kev[ctx->n_kev];
            ev1 = ctx->kev[ctx->n_kev + 1];
            ctx->n_kev += 2;
        }
        if (ctx->cnn_exact_fp32)
            hipExtLaunchKernelGGL(k4_conv12, dim3(nblk), dim3(256), 0, ctx->stream, ev0, ev1, 0, x_batch, packed, a3, nb, scale, scale_mode, site0);
        else
        {
            // the three-stage trunk k5_trunk_p3 is the default since round 5 (-4 % per launch on the bench's tensors); NC_TRUNK_P3=0 selects the
            // two-stage k5_trunk_h3: the same results bit for bit (the same MFMA sequence per accumulator; tests/test_gpu_parity.py)
            const char *e3 = getenv("NC_TRUNK_P3");
            const bool p3 = !(e3 && e3[0] == '0');
            if (trunk_lin_selected(ctx)) {
                // int16 tensors (the product path): conv1 by linearity, k5_trunk_lin (round 6).  Both scale modes are the same arithmetic here: the
                // scale multiplies the accumulators, not the operand
                hipExtLaunchKernelGGL(k5_trunk_lin, dim3(nblk5), dim3(512), 0, ctx->stream, ev0, ev1, 0, reinterpret_cast<const int16_t *>(x_batch), packed_h,
                                      packed_h + H_PACKED_BYTES + FC_PACKED_BYTES, a3, nb, scale, site0, x_limit, ctx->range_sites);
            } else {
                auto *kt = p3 ? (ctx->x_i16 ? k5_trunk_p3<true> : k5_trunk_p3<false>) : (ctx->x_i16 ? k5_trunk_h3<true> : k5_trunk_h3<false>);
                hipExtLaunchKernelGGL(kt, dim3(nblk5), dim3(512), 0, ctx->stream, ev0, ev1, 0, x_batch, packed_h, a3, nb, scale, scale_mode, site0, x_limit, ctx->range_sites);
            }
        }
        if (ctx->cnn_exact_fp32)
            hipLaunchKernelGGL((k3_fc1<F, TMF>), dim3(blocks_for(nb, 16 * TMF)), dim3(256), 0, ctx->stream, a3, (int)n3, kf, bf, f1, nb);
        else
            hipLaunchKernelGGL(k6_fc1_h3, dim3(blocks_for(nb, 16 * FC_TM)), dim3(256), 0, ctx->stream, a3, packed_h + H_PACKED_BYTES, f1, nb);
    } else {
        auto grid = [](int64_t npos) { const int64_t t = (npos + 63) / 64; return dim3((unsigned)(t < 2048 ? t : 2048)); };
        if constexpr (3 * C1 == 24 && C2 == 32 && C3 == 48) {
            if (indel_h3 && !indel_fused) {
                const _Float16 *a2h = reinterpret_cast<const _Float16 *>(a2), *a2l = a2h + np2 * 32;
                hipLaunchKernelGGL((k8_conv23_h3<H2, W2, 32, 48, true>), grid(np3), dim3(256), 0, ctx->stream, a2h, a2l,
                                   packed_h + H3Layer<24, 32>::BYTES, (void *)a3, np2, np3);
            }
        }
        if (!indel_h3) {
            hipLaunchKernelGGL((k7_conv23_mfma<H, W, 3 * C1, C2>), grid(np2), dim3(256), 0, ctx->stream, a1, k2, b2, a2, np2);
            hipLaunchKernelGGL((k7_conv23_mfma<H2, W2, C2, C3>), grid(np3), dim3(256), 0, ctx->stream, a2, k3, b3, a3, np3);
        }
#if defined(NC_FC1_TM) && NC_FC1_TM == 4
        hipLaunchKernelGGL((k3_fc1<F, 4>), dim3(blocks_for(nb, 64)), dim3(256), 0, ctx->stream, a3, (int)n3, kf, bf, f1, nb);
#elif defined(NC_FC1_TM) && NC_FC1_TM == 1
        hipLaunchKernelGGL((k3_fc1<F, 1>), dim3(blocks_for(nb, 16)), dim3(256), 0, ctx->stream, a3, (int)n3, kf, bf, f1, nb);
#else
        hipLaunchKernelGGL((k3_fc1<F, 2>), dim3(blocks_for(nb, 32)), dim3(256), 0, ctx->stream, a3, (int)n3, kf, bf, f1, nb);
#endif
    }
    NC_HIP(ctx, hipGetLastError());
    return NC_OK;
}

}   // namespace

extern "C" {

#if defined(NC_TRACE) || defined(NC_TRACE_BLOCKS) || defined(NC_TRACE_P3)
int nc_debug_trace(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(nc_trace_buf), sizeof(nc_trace_buf)); }
#endif

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
    {
        // Range guard of the split-precision kernels: their epilogues clamp activations at 6e4 (fp16).  With L_l = the largest L1 norm of
        // an output channel's weights and |selu(v)| <= max(lambda |v|, lambda alpha), an input bounded by X bounds every activation:
        //   a_l = max(lambda (L_l a_{l-1} + max|b_l|), lambda alpha),  a_0 = X;    x_limit = the largest X with max a_l < 6e4.
        const bool snp = kind == NC_MODEL_SNP || kind == NC_MODEL_SNP_HAP;
        const int CI = snp ? 5 : 2, C1 = snp ? 16 : 8, C2 = 32, C3 = snp ? 64 : 48;
        auto l1 = [](const float *k, int n_in, int cout) {
            double best = 0;
            for (int c = 0; c < cout; c++) {
                double acc = 0;
                for (int i = 0; i < n_in; i++) acc += std::fabs((double)k[(size_t)i * cout + c]);
                best = std::max(best, acc);
            }
            return best;
        };
        auto amax = [](const float *b, int n) { double m = 0; for (int i = 0; i < n; i++) m = std::max(m, std::fabs((double)b[i])); return m; };
        const float *k11 = blob_host, *b11 = k11 + 5 * CI * C1, *k12 = b11 + C1, *b12 = k12 + 5 * CI * C1, *k13 = b12 + C1, *b13 = k13 + 25 * CI * C1;
        const float *k2 = b13 + C1, *b2 = k2 + 6 * 3 * C1 * C2, *k3 = b2 + C2, *b3 = k3 + 6 * C2 * C3;
        const double L1 = std::max(l1(k11, 5 * CI, C1), std::max(l1(k12, 5 * CI, C1), l1(k13, 25 * CI, C1)));
        const double B1 = std::max(amax(b11, C1), std::max(amax(b12, C1), amax(b13, C1)));
        const double L2 = l1(k2, 6 * 3 * C1, C2), B2 = amax(b2, C2), L3 = l1(k3, 6 * C2, C3), B3 = amax(b3, C3);
        const double LAM = 1.0507009873554805, LA = LAM * 1.6732632423543772, CAP = 60000.0 * 0.999;
        auto worst = [&](double X) {
            const double a1 = std::max(LAM * (L1 * X + B1), LA), a2 = std::max(LAM * (L2 * a1 + B2), LA), a3 = std::max(LAM * (L3 * a2 + B3), LA);
            // the indel kernels clamp conv1's and conv2's outputs only (conv3 leaves k8_conv23_h3 as fp32)
            return snp ? std::max(a1, std::max(a2, a3)) : std::max(a1, a2);
        };
        double lo = 0.0, hi = 1e6;
        if (worst(0.0) >= CAP) hi = 0.0;
        for (int it = 0; it < 60 && hi > 0.0; it++) {
            const double mid = 0.5 * (lo + hi);
            if (worst(mid) < CAP) lo = mid; else hi = mid;
        }
        w.x_limit = (float)lo;
    }
    if (kind == NC_MODEL_SNP || kind == NC_MODEL_SNP_HAP) {
        // B fragments of the fused conv1+conv2 kernel, in (step, lane) order
        std::vector<float> pk((size_t)F12_PACKED, 0.0f);
        const float *k11 = blob_host, *b11 = k11 + 400, *k12 = b11 + 16, *b12 = k12 + 400, *k13 = b12 + 16, *b13 = k13 + 2000;
        const float *k2 = b13 + 16, *b2 = k2 + 2 * 3 * 48 * 32;
        float *w1p = pk.data(), *b1 = w1p + F12_W1P, *w2p = b1 + 48, *b2p = w2p + F12_W2P, *w3p = b2p + 32, *b3p = w3p + F12_W3P;
        const float *k3 = b2 + 32, *b3 = k3 + 2 * 3 * 32 * 64;
        for (int lane = 0; lane < 64; lane++) {
            const int kq = lane >> 4, c = lane & 15;
            for (int s = 0; s < 35; s++) {
                const int dy = s / 7, kl = 4 * (s % 7) + kq;
                if (kl < 25) w1p[s * 64 + lane] = k13[((dy * 5 + kl / 5) * 5 + kl % 5) * 16 + c];
            }
            for (int ls = 0; ls < 7; ls++) {
                const int kl = 4 * ls + kq;
                if (kl < 25) w1p[(35 + ls) * 64 + lane] = k11[kl * 16 + c];
            }
            for (int dy = 0; dy < 5; dy++)
                for (int t = 0; t < 2; t++) {
                    const int kl = 4 * (2 + t) + kq;
                    if (kl >= 10 && kl < 15) w1p[(42 + dy * 2 + t) * 64 + lane] = k12[(dy * 5 + (kl - 10)) * 16 + c];
                }
            for (int tap = 0; tap < 6; tap++)
                for (int j = 0; j < 3; j++)
                    for (int i = 0; i < 4; i++)
                        for (int tn = 0; tn < 2; tn++)
                            w2p[((((tap * 3 + j) * 4 + i) * 2) + tn) * 64 + lane] = k2[(tap * 48 + 16 * j + 4 * kq + i) * 32 + tn * 16 + c];
        }
        for (int c = 0; c < 16; c++) { b1[c] = b11[c]; b1[16 + c] = b12[c]; b1[32 + c] = b13[c]; }
        for (int c = 0; c < 32; c++) b2p[c] = b2[c];
        for (int lane = 0; lane < 64; lane++) {
            const int kq = lane >> 4, c = lane & 15;
            for (int tap = 0; tap < 6; tap++)
                for (int j = 0; j < 2; j++)
                    for (int i = 0; i < 4; i++)
                        for (int tn = 0; tn < 4; tn++)
                            w3p[(((tap * 2 + j) * 4 + i) * 4 + tn) * 64 + lane] = k3[(tap * 32 + 16 * j + 4 * kq + i) * 64 + tn * 16 + c];
        }
        for (int c = 0; c < 64; c++) b3p[c] = b3[c];
        if (!w.packed) {
            hipError_t e = hipMalloc(&w.packed, pk.size() * 4);
            if (e != hipSuccess) return nc_fail(ctx, NC_ERR_NOMEM, "hipMalloc packed weights: %s", hipGetErrorString(e));
        }
        NC_HIP(ctx, hipMemcpyAsync(w.packed, pk.data(), pk.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        w.n_packed = pk.size();
        // ---- fp16x3 fragments: weights scaled by a power of two S (so that |w| * S <= 16384), split hi/lo on the host
        float wmax = 0.0f;
        for (const float *q = blob_host; q < b3 + 64; q++) wmax = std::fmax(wmax, std::fabs(*q));
        float S = 1024.0f;
        while (S > 1.0f && wmax * S > 16384.0f) S *= 0.5f;
        std::vector<uint8_t> hp((size_t)H_PACKED_BYTES + FC_PACKED_BYTES + L_PACKED_BYTES, 0);
        _Float16 *w1f = reinterpret_cast<_Float16 *>(hp.data()), *w2h = w1f + T_NW1 * T_FRAG, *w2l = w2h + T_NW2 * T_FRAG,
                 *w3h = w2l + T_NW2 * T_FRAG, *w3l = w3h + T_NW3 * T_FRAG;
        float *b1s = reinterpret_cast<float *>(w3l + T_NW3 * T_FRAG), *b2s = b1s + 48, *b3s = b2s + 32;
        int32_t *c3tab = reinterpret_cast<int32_t *>(b3s + 68);
        auto split = [&](float v, _Float16 &h, _Float16 &l) {
            const float sv = v * S;
            h = (_Float16)sv;
            l = (_Float16)(sv - (float)h);
        };
        // conv1 fragment pair (WA at fragment fa, WB at fragment fb) of one lane: kw[ci] = the 5 channel weights of its tap
        auto put_tap = [&](int fa, int fb, int lane, const float *kw /* 5 or nullptr */) {
            _Float16 H[5], L[5];
            for (int ci = 0; ci < 5; ci++) split(kw ? kw[ci] : 0.0f, H[ci], L[ci]);
            _Float16 *A = w1f + ((size_t)fa * 64 + lane) * 8, *B = w1f + ((size_t)fb * 64 + lane) * 8;
            const _Float16 wa[8] = {H[0], H[1], H[2], H[3], H[4], H[0], H[1], H[2]};
            const _Float16 wb[8] = {H[3], H[4], L[0], L[1], L[2], L[3], L[4], (_Float16)0.0f};
            for (int j = 0; j < 8; j++) { A[j] = wa[j]; B[j] = wb[j]; }
        };
        for (int lane = 0; lane < 64; lane++) {
            const int g = lane >> 4, c = lane & 15;
            for (int G = 0; G < 7; G++) {
                const int dy = C1_TAPS[G][g][0], dx = C1_TAPS[G][g][1];
                // a tap listed twice in a group (dummy lanes) carries its weights only at its first occurrence
                bool first = true;
                for (int q = 0; q < g; q++) first = first && !(C1_TAPS[G][q][0] == dy && C1_TAPS[G][q][1] == dx);
                float kw[5];
                for (int ci = 0; ci < 5; ci++) kw[ci] = k13[((dy * 5 + dx) * 5 + ci) * 16 + c];
                put_tap(G, 7 + G, lane, first ? kw : nullptr);
                if (G < 2) {                                                         // 1x5 kernel: taps of image row dy == 2
                    for (int ci = 0; ci < 5; ci++) kw[ci] = k11[(dx * 5 + ci) * 16 + c];
                    put_tap(14 + G, 16 + G, lane, (first && dy == 2) ? kw : nullptr);
                }
                if (G >= 1 && G <= 3) {                                              // 5x1 kernel: taps of image column dx == 2
                    for (int ci = 0; ci < 5; ci++) kw[ci] = k12[(dy * 5 + ci) * 16 + c];
                    put_tap(18 + G - 1, 21 + G - 1, lane, (first && dx == 2) ? kw : nullptr);
                }
            }
            for (int j = 0; j < 8; j++) {
                for (int G = 0; G < 9; G++) {
                    const int idx = 4 * G + g, tap = idx / 6, ch = (idx % 6) * 8 + j;
                    for (int tn = 0; tn < 2; tn++) {
                        const size_t o = ((size_t)(G * 2 + tn) * 64 + lane) * 8 + j;
                        split(k2[(tap * 48 + ch) * 32 + tn * 16 + c], w2h[o], w2l[o]);
                    }
                }
                for (int G = 0; G < 6; G++)
                    for (int tn = 0; tn < 4; tn++) {
                        const size_t o = ((size_t)(G * 4 + tn) * 64 + lane) * 8 + j;
                        split(k3[(G * 32 + 8 * g + j) * 64 + tn * 16 + c], w3h[o], w3l[o]);
                    }
            }
        }
        for (int c = 0; c < 16; c++) { b1s[c] = b11[c] * S; b1s[16 + c] = b12[c] * S; b1s[32 + c] = b13[c] * S; }
        for (int c = 0; c < 32; c++) b2s[c] = b2[c] * S;
        for (int c = 0; c < 64; c++) b3s[c] = b3[c] * S;
        b3s[64] = 1.0f / S;
        for (int i = 0; i < 32; i++) { c3tab[i] = C3_SLOT[i]; c3tab[32 + i] = C3_OUT[i]; }
        {
            // fc1 fragments for k6_fc1_h3: own power-of-two scale, A operand = W^T (row = output unit, K = 32 G + 8 g + j)
            const float *kf = b3 + 64, *bf = kf + 1728 * 48;
            float fmax_ = 0.0f;
            for (const float *q = kf; q < bf + 48; q++) fmax_ = std::fmax(fmax_, std::fabs(*q));
            float SF = 1024.0f;
            while (SF > 1.0f && fmax_ * SF > 16384.0f) SF *= 0.5f;
            _Float16 *fh = reinterpret_cast<_Float16 *>(hp.data() + H_PACKED_BYTES), *fl = fh + (size_t)FC_G * FC_TN * T_FRAG;
            float *fbs = reinterpret_cast<float *>(fl + (size_t)FC_G * FC_TN * T_FRAG);
            for (int G = 0; G < FC_G; G++)
                for (int tn = 0; tn < FC_TN; tn++)
                    for (int lane = 0; lane < 64; lane++)
                        for (int j = 0; j < 8; j++) {
                            const int g = lane >> 4, c = lane & 15;
                            const float sv = kf[(size_t)(32 * G + 8 * g + j) * 48 + tn * 16 + c] * SF;
                            const _Float16 h = (_Float16)sv;
                            const size_t o = ((size_t)(G * FC_TN + tn) * 64 + lane) * 8 + j;
                            fh[o] = h;
                            fl[o] = (_Float16)(sv - (float)h);
                        }
            for (int c = 0; c < 48; c++) fbs[c] = bf[c] * SF;
            fbs[48] = 1.0f / SF;
        }
        {
            // conv1 fragments of k5_trunk_lin (record form): K slot k = 6 i + v of a 5-tap kernel row = pixel i (column offset L_DXO[i]), value v of
            // [x0 x1 x2 x3 hi(u4 rho) lo(u4 rho)]; the lo-weight fragment has no entry for v = 5 (lo x lo is dropped, as everywhere)
            _Float16 *lf = reinterpret_cast<_Float16 *>(hp.data() + H_PACKED_BYTES + FC_PACKED_BYTES);
            auto lput = [&](int frag, int lane, int j, _Float16 v) { lf[((size_t)frag * 64 + lane) * 8 + j] = v; };
            const _Float16 Z = (_Float16)0.0f;
            for (int lane = 0; lane < 64; lane++) {
                const int g = lane >> 4, c = lane & 15;
                for (int j = 0; j < 8; j++) {
                    _Float16 H, L;
                    const int k = 8 * g + j;
                    if (k < 30) {
                        const int i = k / 6, v = k % 6, dx = 2 + L_DXO[i], ci = v < 4 ? v : 4;
                        for (int dy = 0; dy < 5; dy++) {
                            split(k13[((dy * 5 + dx) * 5 + ci) * 16 + c], H, L);
                            lput(LW_5H + dy, lane, j, H);
                            lput(LW_5L + dy, lane, j, v == 5 ? Z : L);
                        }
                        split(k11[(dx * 5 + ci) * 16 + c], H, L);
                        lput(LW_1H, lane, j, H);
                        lput(LW_1L, lane, j, v == 5 ? Z : L);
                    }
                    // 5x1 kernel: slot 0 of a record = the centre pixel's six values (+ two of another pixel: zero weights)
                    if (j < 6) {
                        const int ci = j < 4 ? j : 4;
                        split(k12[(g * 5 + ci) * 16 + c], H, L);                       // group A: kernel row dy = g, rows 1..4
                        lput(LW_2HA, lane, j, H);
                        lput(LW_2LA, lane, j, j == 5 ? Z : L);
                        const int dyb = g == 0 ? 4 : g - 1;                             // group B: g = 0: dy = 4 of rows 1..4; g >= 1: dy = g - 1 of the row-0 hi plane
                        split(k12[(dyb * 5 + ci) * 16 + c], H, L);
                        lput(LW_2HB, lane, j, (g > 0 && j == 5) ? Z : H);
                        lput(LW_2LB, lane, j, j == 5 ? Z : L);
                        if (g < 3) {                                                     // group C: dy = g of the row-0 lo plane
                            split(k12[(g * 5 + ci) * 16 + c], H, L);
                            lput(LW_2HC, lane, j, j == 5 ? Z : H);
                        }
                    }
                }
            }
        }
        if (!w.packed_h) {
            hipError_t e = hipMalloc(&w.packed_h, hp.size());
            if (e != hipSuccess) return nc_fail(ctx, NC_ERR_NOMEM, "hipMalloc packed fp16 weights: %s", hipGetErrorString(e));
        }
        NC_HIP(ctx, hipMemcpyAsync(w.packed_h, hp.data(), hp.size(), hipMemcpyHostToDevice, ctx->stream));
        NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        // indel models: split-precision fragments of conv2 and conv3 (k8_conv23_h3); conv1 (CI = 2) and fc1 stay fp32
        const float *k2 = blob_host + (5 + 5 + 25) * 2 * 8 + 3 * 8, *b2 = k2 + 6 * 24 * 32, *k3 = b2 + 32, *b3 = k3 + 6 * 32 * 48;
        std::vector<uint8_t> hp(INDEL_H3_BYTES, 0);
        pack_h3_layer<24, 32>(k2, b2, hp.data());
        pack_h3_layer<32, 48>(k3, b3, hp.data() + H3Layer<24, 32>::BYTES);
        pack_conv1_h3(blob_host, hp.data() + H3Layer<24, 32>::BYTES + H3Layer<32, 48>::BYTES);
        if (!w.packed_h) {
            hipError_t e = hipMalloc(&w.packed_h, hp.size());
            if (e != hipSuccess) return nc_fail(ctx, NC_ERR_NOMEM, "hipMalloc packed fp16 weights: %s", hipGetErrorString(e));
        }
        NC_HIP(ctx, hipMemcpyAsync(w.packed_h, hp.data(), hp.size(), hipMemcpyHostToDevice, ctx->stream));
        NC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return NC_OK;
}

int nc_snp_forward(nc_ctx *ctx, int32_t kind, int64_t n, const float *x_dev, const int32_t *ref_code_dev, const double *scale_dev,
                   int32_t scale_mode, float *probs_dev, float *gt_dev)
{
    return nc_snp_forward_drain(ctx, kind, n, x_dev, ref_code_dev, scale_dev, scale_mode, probs_dev, gt_dev, nullptr, nullptr, nullptr);
}

int nc_snp_forward_drain(nc_ctx *ctx, int32_t kind, int64_t n, const float *x_dev, const int32_t *ref_code_dev, const double *scale_dev,
                         int32_t scale_mode, float *probs_dev, float *gt_dev, void *copy_stream, float *probs_host, float *gt_host)
{
    if (!ctx) return NC_ERR_ARG;
    if (probs_host && !copy_stream) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_forward_drain: host drain needs a copy stream");
    if (kind != NC_MODEL_SNP && kind != NC_MODEL_SNP_HAP) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_forward: not an SNP model kind");
    if (!ctx->w[kind].dev) return nc_fail(ctx, NC_ERR_STATE, "nc_snp_forward: weights of kind %d not loaded", kind);
    if (n < 0 || (n && (!x_dev || !ref_code_dev || !probs_dev))) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_forward: null argument");
    if (scale_mode != 0 && scale_mode != 1) return nc_fail(ctx, NC_ERR_ARG, "nc_snp_forward: scale_mode");
    if (ctx->x_i16 && ctx->cnn_exact_fp32) return nc_fail(ctx, NC_ERR_STATE, "nc_snp_forward: int16 tensors are read by the split-precision trunk only");
    NC_HIP(ctx, hipSetDevice(ctx->device));
    NcTimer tm(ctx, 2);
    if (ctx->timing) nc_timing_resolve(ctx, 4);   // fold an earlier call's per-launch events in before they are re-used
    ctx->n_kev = 0;
    // 262,144 sites per batch (1.8 GB of conv3 activations): the kernel boundaries between trunk, fc1 and heads cost ~40 us per
    // batch, and a batch still drains to the host while the next one computes; one launch for a whole 625k-site contig is
    // slower again (the trunk itself loses 3 % on a 4.3 GB activation buffer)
    const int64_t BATCH = 262144;
    for (int64_t s0 = 0; s0 < n; s0 += BATCH) {
        const int64_t nb = n - s0 < BATCH ? n - s0 : BATCH;
        const float *tail = nullptr, *f1 = nullptr;
        NC_TRY((run_trunk<5, 41, 5, 16, 32, 64, 48, 2, 1, true>(ctx, ctx->w[kind].dev, ctx->w[kind].packed, (const uint8_t *)ctx->w[kind].packed_h, s0, nb, ctx->x_i16 ? (const float *)((const int16_t *)x_dev + s0 * NC_SNP_TENSOR) : x_dev + s0 * NC_SNP_TENSOR, scale_dev, scale_mode,
                                                          &f1, &tail, ctx->w[kind].x_limit)));
        if (kind == NC_MODEL_SNP)
            hipLaunchKernelGGL(k_snp_heads, dim3(blocks_for(nb)), dim3(256), 0, ctx->stream, f1, tail, ref_code_dev + s0, nb,
                               probs_dev + s0 * 4, gt_dev ? gt_dev + s0 * 2 : nullptr);
        else
            hipLaunchKernelGGL(k_snp_hap_heads, dim3(blocks_for(nb)), dim3(256), 0, ctx->stream, f1, tail, ref_code_dev + s0, nb,
                               probs_dev + s0 * 4);
        NC_HIP(ctx, hipGetLastError());
        if (probs_host) {                    // drain this batch on the copy stream while the next batch computes
            const int slot = (int)((s0 / BATCH) & 3);
            if (!ctx->drain_ev[slot]) NC_HIP(ctx, hipEventCreateWithFlags(&ctx->drain_ev[slot], hipEventDisableTiming));
            NC_HIP(ctx, hipEventRecord(ctx->drain_ev[slot], ctx->stream));
            NC_HIP(ctx, hipStreamWaitEvent((hipStream_t)copy_stream, ctx->drain_ev[slot], 0));
            NC_TRY(nc_d2h(ctx, probs_host + s0 * 4, probs_dev + s0 * 4, (size_t)nb * 16, (hipStream_t)copy_stream));
            if (gt_host && gt_dev)
                NC_TRY(nc_d2h(ctx, gt_host + s0 * 2, gt_dev + s0 * 2, (size_t)nb * 8, (hipStream_t)copy_stream));
        }
    }
    tm.stop();
    if (ctx->timing) ctx->kev_pending = true;     // resolved by nc_last_kernel_ms(4 / 5)
    return NC_OK;
}

int nc_cnn_x_limit(nc_ctx *ctx, int32_t kind, float *x_limit)
{
    if (!ctx || kind < 0 || kind > 3 || !x_limit) return NC_ERR_ARG;
    if (!ctx->w[kind].dev) return nc_fail(ctx, NC_ERR_STATE, "nc_cnn_x_limit: weights of kind %d not loaded", kind);
    *x_limit = ctx->w[kind].x_limit;
    return NC_OK;
}

int nc_snp_trunk_info(nc_ctx *ctx, int32_t *mfma_per_site, int32_t *kernel_id)
{
    if (!ctx) return NC_ERR_ARG;
    const char *e3 = getenv("NC_TRUNK_P3");
    const bool lin = trunk_lin_selected(ctx);
    if (mfma_per_site) *mfma_per_site = lin ? L_MFMA_PER_SITE : 13 * 24 + 10 * 27 + 8 * 18;
    if (kernel_id) *kernel_id = ctx->cnn_exact_fp32 ? 0 : lin ? 3 : (e3 && e3[0] == '0') ? 1 : 2;
    return NC_OK;
}

int nc_cnn_range_watch(nc_ctx *ctx, uint8_t *site_flags_dev)
{
    if (!ctx) return NC_ERR_ARG;
    ctx->range_sites = site_flags_dev;
    return NC_OK;
}

int nc_indel_forward(nc_ctx *ctx, int32_t kind, int64_t n, const float *x_dev, float *probs_dev)
{
    if (!ctx) return NC_ERR_ARG;
    if (kind != NC_MODEL_INDEL && kind != NC_MODEL_INDEL_HAP) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_forward: not an indel model kind");
    if (!ctx->w[kind].dev) return nc_fail(ctx, NC_ERR_STATE, "nc_indel_forward: weights of kind %d not loaded", kind);
    if (n < 0 || (n && (!x_dev || !probs_dev))) return nc_fail(ctx, NC_ERR_ARG, "nc_indel_forward: null argument");
    NC_HIP(ctx, hipSetDevice(ctx->device));
    const int nout = kind == NC_MODEL_INDEL ? 4 : 1;
    const int64_t xs = kind == NC_MODEL_INDEL ? 15 * 128 * 2 : 5 * 128 * 2;
    NcTimer tm(ctx, 2);
    // batch = what the conv3 activations (fc1's input, 77 KB / 18 KB per site) may take in HBM: ~5 GB.  NC_INDEL_TRUNK_SPLIT's kernels keep conv2's too
    static const bool split_env = getenv("NC_INDEL_TRUNK_SPLIT") != nullptr;
    int64_t BATCH = split_env ? (kind == NC_MODEL_INDEL ? 16384 : 32768) : (kind == NC_MODEL_INDEL ? 65536 : 262144);
    if (const char *be = getenv("NC_INDEL_BATCH")) BATCH = std::max<int64_t>(256, atoll(be));      // (experiments: conv3's activations inside the memory-side cache)
    for (int64_t s0 = 0; s0 < n; s0 += BATCH) {
        const int64_t nb = n - s0 < BATCH ? n - s0 : BATCH;
        const float *tail = nullptr, *f1 = nullptr;
        // msa() tensors are frequencies: |x| <= 1.  A model whose L1 norms do not prove the fp16 range safe for such inputs (x_limit < 1;
        // none of the shipped ones) runs on the exact fp32 kernels instead
        const uint8_t *ph = ctx->w[kind].x_limit >= 1.0f ? (const uint8_t *)ctx->w[kind].packed_h : nullptr;
        if (kind == NC_MODEL_INDEL)
            NC_TRY((run_trunk<15, 128, 2, 8, 32, 48, 32, 2, 1, false>(ctx, ctx->w[kind].dev, nullptr, ph, s0, nb, x_dev + s0 * xs, nullptr, 0, &f1, &tail)));
        else
            NC_TRY((run_trunk<5, 128, 2, 8, 32, 48, 32, 2, 1, false>(ctx, ctx->w[kind].dev, nullptr, ph, s0, nb, x_dev + s0 * xs, nullptr, 0, &f1, &tail)));
        hipLaunchKernelGGL(k_indel_heads, dim3(blocks_for(nb)), dim3(256), 0, ctx->stream, f1, tail, nout, nb, probs_dev + s0 * nout);
        NC_HIP(ctx, hipGetLastError());
    }
    tm.stop();
    return NC_OK;
}

}   // namespace
