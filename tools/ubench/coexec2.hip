// Microbenchmark (experiment): which instruction classes overlap with v_mfma_f32_16x16x32_f16 issued by the OTHER wave of the
// same SIMD on gfx950?  Workgroup = 8 waves (2 per SIMD): waves 0-3 run MFMAs, waves 4-7 run the filler op.  Three timings per
// op: MFMA waves only, filler waves only, both; overlap = (Tm + Tf - Tboth) / min(Tm, Tf).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

enum { OP_FMA, OP_PKFMA, OP_EXP, OP_CVT, OP_MIX, OP_MED3, OP_DSW, OP_DSR, OP_MUL, OP_N };
static const char *OP_NAME[OP_N] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "v_med3_f32", "ds_write_b64", "ds_read_b128", "v_mul_f32"};

template <int OP>
__device__ __forceinline__ void filler(float (&v)[8], float (&p)[8], _Float16 *lds, int lane)
{
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(p[0]));
        if (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(p[0]));
        if (OP == OP_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double *>(&v[j & 6])) : "v"(*reinterpret_cast<double *>(&p[0])));
        if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        if (OP == OP_CVT) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v[j]) : "v"(p[j]), "v"(p[(j + 1) & 7]));
        if (OP == OP_MIX) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[j]) : "v"(p[j]));
        if (OP == OP_MED3) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(v[j]) : "v"(p[0]));
        if (OP == OP_DSW) asm volatile("ds_write_b64 %0, %1" ::"v"(lane * 8 + j * 512), "v"(*reinterpret_cast<double *>(&p[j & 6])) : "memory");
        if (OP == OP_DSR) {
            f4 t;
            asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lane * 16 + j * 1024) : "memory");
            if (j == 7) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); v[0] += t[0]; }
        }
    }
}

template <int OP, int BIG>
__global__ __launch_bounds__(512) void k(float *out, int iters, int mode)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[16384];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    h8 w, x;
    for (int j = 0; j < 8; j++) { w[j] = (_Float16)(0.01f * (j + lane % 3)); x[j] = (_Float16)(0.02f * (j + lane % 5)); }
    float v[8], p[8];
    for (int j = 0; j < 8; j++) { v[j] = 0.001f * (lane + j); p[j] = 1.0f + 0.0001f * j; }
    for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (_Float16)0.0f;
    __syncthreads();
    float s = 0;
    if (wv < 4) {
        if (mode & 1) {
            if (BIG) {
                f16v acc[2] = {};
                for (int it = 0; it < iters; it++) {
#pragma unroll
                    for (int m = 0; m < 8; m++) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, acc[m & 1], 0, 0, 0);
                }
                s += acc[0][0] + acc[1][3];
            } else {
                f4 acc[4] = {};
                for (int it = 0; it < iters; it++) {
#pragma unroll
                    for (int m = 0; m < 16; m++) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, acc[m & 3], 0, 0, 0);
                }
                s += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
            }
        }
    } else if (mode & 2) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 6; r++) filler<OP>(v, p, lds, lane);            // 48 filler ops per 16 MFMAs: 3 per MFMA
        }
        for (int j = 0; j < 8; j++) s += v[j];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)lds[threadIdx.x];
}

template <int OP, int BIG>
float time_mode(float *out, int mode)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, BIG>), dim3(256), dim3(512), 0, 0, out, 50, mode);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, BIG>), dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;                                                   // ns per iteration
}

template <int OP, int BIG>
void run(float *out)
{
    const float tm = time_mode<OP, BIG>(out, 1), tf = time_mode<OP, BIG>(out, 2), tb = time_mode<OP, BIG>(out, 3);
    const float lo = tm < tf ? tm : tf;
    printf("%-18s mfma=%s: mfma-only %6.1f ns/iter (%.1f ns per 16x16x32-equivalent), filler-only %6.1f (%.2f ns per op), both %6.1f, overlap %.2f\n",
           OP_NAME[OP], BIG ? "32x32x16" : "16x16x32", tm, tm / 16, tf, tf / 48, tb, (tm + tf - tb) / lo);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    run<OP_FMA, 0>(out); run<OP_MUL, 0>(out); run<OP_PKFMA, 0>(out); run<OP_EXP, 0>(out); run<OP_CVT, 0>(out); run<OP_MIX, 0>(out);
    run<OP_MED3, 0>(out); run<OP_DSW, 0>(out); run<OP_DSR, 0>(out);
    run<OP_FMA, 1>(out); run<OP_EXP, 1>(out); run<OP_DSR, 1>(out);
    hipFree(out);
    return 0;
}
