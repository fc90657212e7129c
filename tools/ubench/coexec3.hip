// Microbenchmark (experiment): does VALU work of the OTHER wave of a SIMD overlap with v_mfma_f32_16x16x32_f16 when the MFMA wave does
// not sit at the issue port with its next MFMA?  coexec2.hip found no overlap with MFMAs issued back to back; here the MFMA wave puts
// `s_nop N` (or a ds_read_b128, or an s_setprio pair) after every MFMA.  Workgroup = 8 waves (2 per SIMD): waves 0-3 MFMA, 4-7 v_fma_f32.
// Three timings per variant: MFMA waves only, filler waves only, both; overlap = (Tm + Tf - Tboth) / min(Tm, Tf).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// GAP: 0 none, 1..8 = s_nop (GAP-1), 9 = one ds_read_b128 per MFMA, 10 = s_setprio 0 on the MFMA waves / 3 on the fillers,
// 11 = s_sleep 0 ... ; NACC accumulators in rotation
template <int GAP>
__device__ __forceinline__ void gap(int lane, int m)
{
    if (GAP == 1) asm volatile("s_nop 0");
    if (GAP == 2) asm volatile("s_nop 1");
    if (GAP == 3) asm volatile("s_nop 2");
    if (GAP == 4) asm volatile("s_nop 3");
    if (GAP == 5) asm volatile("s_nop 5");
    if (GAP == 6) asm volatile("s_nop 7");
    if (GAP == 7) asm volatile("s_nop 9");
    if (GAP == 8) asm volatile("s_nop 11");
    if (GAP == 9) { f4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lane * 16 + (m & 7) * 1024) : "memory"); }
    if (GAP == 11) asm volatile("s_sleep 0");
    if (GAP == 12) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0");
}

template <int GAP, int NF, int NACC = 4, int SWAP = 0, int PRIO = 0, int FOP = 0>
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
    if ((wv < 4) != (SWAP != 0)) {
        if (mode & 1) {
            if (GAP == 10) asm volatile("s_setprio 0");
            if (PRIO == 1) asm volatile("s_setprio 3");
            if (PRIO == 2) asm volatile("s_setprio 0");
            f4 acc[4] = {};
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int m = 0; m < 16; m++) {
                    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(w), "v"(x));
                    gap<GAP>(lane, m);
                }
                if (GAP == 9) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            s += acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
        }
    } else if (mode & 2) {
        if (GAP == 10) asm volatile("s_setprio 3");
        if (PRIO == 2) asm volatile("s_setprio 3");
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < NF * 2; r++)
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    if (FOP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[j]) : "v"(p[0]));          // NF VALU per MFMA
                    if (FOP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double *>(&v[j & 6])) : "v"(*reinterpret_cast<double *>(&p[0])));
                    if (FOP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
                    if (FOP == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(v[j]) : "v"(p[j]), "v"(p[(j + 1) & 7]));
                    if (FOP == 4) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v[j]) : "v"(p[j]));
                    if (FOP == 5) asm volatile("v_med3_f32 %0, %0, 0, %1" : "+v"(v[j]) : "v"(p[0]));
                    if (FOP == 6) asm volatile("ds_write_b64 %0, %1" ::"v"(lane * 8 + j * 512), "v"(*reinterpret_cast<double *>(&p[j & 6])) : "memory");
                    if (FOP == 7) { f4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lane * 16 + j * 1024) : "memory"); if (j == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
                    if (FOP == 8) asm volatile("v_exp_f32_e64 %0, %0 clamp" : "+v"(v[j]));
                    if (FOP == 9) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*reinterpret_cast<double *>(&v[j & 6])) : "v"(*reinterpret_cast<double *>(&p[0])));
                }
        }
        for (int j = 0; j < 8; j++) s += v[j];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)lds[threadIdx.x];
}

template <int GAP, int NF, int NACC = 4, int SWAP = 0, int PRIO = 0, int FOP = 0>
float time_mode(float *out, int mode)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<GAP, NF, NACC, SWAP, PRIO, FOP>), dim3(256), dim3(512), 0, 0, out, 50, mode);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<GAP, NF, NACC, SWAP, PRIO, FOP>), dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters;
}

template <int GAP, int NF, int NACC = 4, int SWAP = 0, int PRIO = 0, int FOP = 0>
void run(float *out, const char *name)
{
    const float tm = time_mode<GAP, NF, NACC, SWAP, PRIO, FOP>(out, 1), tf = time_mode<GAP, NF, NACC, SWAP, PRIO, FOP>(out, 2), tb = time_mode<GAP, NF, NACC, SWAP, PRIO, FOP>(out, 3);
    const float lo = tm < tf ? tm : tf;
    printf("%-28s %d v_fma per MFMA in the other wave: mfma-only %6.1f ns per 16 (%.2f each), filler-only %6.1f, both %6.1f, overlap %.2f\n", name, NF, tm, tm / 16,
           tf, tb, (tm + tf - tb) / lo);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    run<2, 2, 4, 0, 0, 0>(out, "nop1 older MFMA | v_fma_f32");
    run<2, 2, 4, 0, 0, 1>(out, "nop1 older MFMA | v_pk_fma_f32");
    run<2, 2, 4, 0, 0, 2>(out, "nop1 older MFMA | v_exp_f32");
    run<2, 2, 4, 0, 0, 8>(out, "nop1 older MFMA | v_exp clamp");
    run<2, 2, 4, 0, 0, 3>(out, "nop1 older MFMA | v_cvt_pk_f16");
    run<2, 2, 4, 0, 0, 4>(out, "nop1 older MFMA | v_fma_mix");
    run<2, 2, 4, 0, 0, 5>(out, "nop1 older MFMA | v_med3");
    run<2, 2, 4, 0, 0, 6>(out, "nop1 older MFMA | ds_write_b64");
    run<2, 2, 4, 0, 0, 7>(out, "nop1 older MFMA | ds_read_b128");
    run<2, 2, 4, 0, 0, 9>(out, "nop1 older MFMA | v_mul_f64");
    run<0, 2, 4, 1, 0, 1>(out, "b2b younger MFMA | v_pk_fma_f32");
    run<0, 2, 4, 1, 0, 2>(out, "b2b younger MFMA | v_exp_f32");
    run<0, 2, 4, 1, 0, 3>(out, "b2b younger MFMA | v_cvt_pk_f16");
    run<0, 2, 4, 1, 0, 7>(out, "b2b younger MFMA | ds_read_b128");
    run<0, 1, 4, 1, 0, 7>(out, "b2b younger MFMA | ds_read_b128 x1");
    run<2, 1, 4, 0, 0, 7>(out, "nop1 older MFMA | ds_read_b128 x1");
    hipFree(out);
    return 0;
}
