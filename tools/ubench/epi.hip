// Microbenchmark (experiment): does the trunk's REAL epilogue (SELU + fp16 hi / lo split + ds_write_b64 of NA accumulators) in one wave overlap with
// v_mfma_f32_16x16x32_f16 (+ s_nop 1) in the other wave of the SIMD?  VAR: 0 the product's code (compiler packs v_pk_fma_f32), 1 plain v_fma_f32 by
// inline asm, 2 + the lo halves by v_fma_mixlo/hi_f16 instead of v_fma_mix_f32 + v_cvt_pk_f16_f32.  OLD: the MFMA waves are the older ones (0-3).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define SELU_L 1.0507009873554805f
#define SELU_LA 1.7580993408473766f
struct h_epi { float c1, c2, c3; };
__device__ __forceinline__ float exp2_clamp01(float x) { float r; asm("v_exp_f32_e64 %0, %1 clamp" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ float sub_h_lo(float v, uint32_t hpk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v)); return r; }
__device__ __forceinline__ float sub_h_hi(float v, uint32_t hpk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v)); return r; }
__device__ __forceinline__ float fma_s(float a, float b, float c) { float r; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int VAR>
__device__ __forceinline__ f4 selu4(const f4 &acc, const h_epi &k, float la, float nla)
{
    f4 s;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float a = acc[r];
        const float e = exp2_clamp01(a * k.c1);
        const float pos = __builtin_amdgcn_fmed3f(a, 0.0f, k.c3);
        if (VAR == 0) { const float neg = fmaf(e, SELU_LA, -SELU_LA); s[r] = fmaf(pos, k.c2, neg); }
        else { const float neg = fma_s(e, la, nla); s[r] = fma_s(pos, k.c2, neg); }
    }
    return s;
}
template <int VAR>
__device__ __forceinline__ void split4_store(const f4 &v, _Float16 *hp, _Float16 *lp)
{
    const h2 h01 = __builtin_convertvector((f2){v[0], v[1]}, h2), h23 = __builtin_convertvector((f2){v[2], v[3]}, h2);
    const uint32_t u01 = __builtin_bit_cast(uint32_t, h01), u23 = __builtin_bit_cast(uint32_t, h23);
    uint32_t l01, l23;
    if (VAR < 2) {
        const f2 d01 = {sub_h_lo(v[0], u01), sub_h_hi(v[1], u01)}, d23 = {sub_h_lo(v[2], u23), sub_h_hi(v[3], u23)};
        l01 = __builtin_bit_cast(uint32_t, __builtin_convertvector(d01, h2));
        l23 = __builtin_bit_cast(uint32_t, __builtin_convertvector(d23, h2));
    } else {
        // lo = f16(v - f32(hi)) in one instruction per value: v_fma_mixlo_f16 / v_fma_mixhi_f16 (f16 source read in place, f16 result written in place)
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(u01), "v"(v[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(u01), "v"(v[1]));
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(u23), "v"(v[2]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(u23), "v"(v[3]));
    }
    *reinterpret_cast<uint2 *>(hp) = make_uint2(u01, u23);
    *reinterpret_cast<uint2 *>(lp) = make_uint2(l01, l23);
}

template <int VAR, int NA, int OLD>
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode, float c1, float c2, float c3, float la)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[32768];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    h8 w, x;
    for (int j = 0; j < 8; j++) { w[j] = (_Float16)(0.01f * (j + lane % 3)); x[j] = (_Float16)(0.02f * (j + lane % 5)); }
    for (int i = threadIdx.x; i < 32768; i += 512) lds[i] = (_Float16)0.0f;
    __syncthreads();
    const h_epi e = {c1, c2, c3};
    float s = 0;
    if ((wv < 4) == (OLD != 0)) {
        if (mode & 1) {
            f4 acc[4] = {};
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int m = 0; m < 16; m++) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 1" : "+v"(acc[m & 3]) : "v"(w), "v"(x));
            }
            s += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        }
    } else if (mode & 2) {
        f4 a[NA];
        for (int q = 0; q < NA; q++) a[q] = (f4){0.1f * lane - 3.0f + q, 0.2f * lane - 5.0f, 0.01f * q, -0.3f * lane};
        _Float16 *hp = lds + (wv & 3) * 4096 + lane * 4, *lp = hp + 16384;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int q = 0; q < NA; q++) {
                asm volatile("" : "+v"(a[q]));
                split4_store<VAR>(selu4<VAR>(a[q], e, la, -la), hp + q * 256, lp + q * 256);
            }
        }
        s += a[0][0];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)lds[threadIdx.x];
}

template <int VAR, int NA, int OLD>
void run(float *out)
{
    float t[4];
    for (int mode = 1; mode <= 3; mode++) {
        const int iters = 2000;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<VAR, NA, OLD>), dim3(256), dim3(512), 0, 0, out, 50, mode, 0.01f, 0.02f, 1000.0f, SELU_LA);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<VAR, NA, OLD>), dim3(256), dim3(512), 0, 0, out, iters, mode, 0.01f, 0.02f, 1000.0f, SELU_LA);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        t[mode] = ms * 1e6f / iters;
    }
    const float lo = t[1] < t[2] ? t[1] : t[2];
    printf("epilogue variant %d, %d accumulators per 16 MFMAs, MFMA waves %s: mfma-only %6.1f ns, epilogue-only %6.1f (%.1f ns per activation), both %6.1f, overlap %.2f\n", VAR, NA,
           OLD ? "older" : "younger", t[1], t[2], t[2] / (4 * NA), t[3], (t[1] + t[2] - t[3]) / lo);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    run<0, 2, 1>(out); run<1, 2, 1>(out); run<2, 2, 1>(out);
    run<0, 2, 0>(out); run<1, 2, 0>(out); run<2, 2, 0>(out);
    run<0, 4, 1>(out); run<1, 4, 1>(out); run<2, 4, 1>(out);
    run<0, 1, 1>(out); run<2, 1, 1>(out);
    hipFree(out);
    return 0;
}
