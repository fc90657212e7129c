// Microbenchmark (experiment): can VALU work hide in the shadow of v_mfma_f32_16x16x32_f16 on gfx950?
//  same-wave: each MFMA followed by F independent v_fma_f32; cross-wave: waves 0-3 MFMA only, waves 4-7 VALU only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int F, int NACC, bool TRANS>
__device__ __forceinline__ void body(f4 (&acc)[NACC], float (&v)[8], const h8 &w, const h8 &x, int iters)
{
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 12; m++) {
            acc[m % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, acc[m % NACC], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < F; f++) {
                if (TRANS && f == 0) v[(m + f) % 8] = __builtin_amdgcn_exp2f(v[(m + f) % 8]);
                else v[(m * F + f) % 8] = __builtin_fmaf(v[(m * F + f) % 8], 1.0001f, 0.5f);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ROLE 0: all waves same-wave interleave.  ROLE 1: waves < NW/2 MFMA only, others VALU only (same total work as F fillers per MFMA)
template <int F, int NACC, int NW, int ROLE, bool TRANS>
__global__ __launch_bounds__(NW * 64) void k(float *out, int iters)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    h8 w, x;
    for (int j = 0; j < 8; j++) { w[j] = (_Float16)(0.01f * (j + lane % 3)); x[j] = (_Float16)(0.02f * (j + lane % 5)); }
    f4 acc[NACC];
    for (int t = 0; t < NACC; t++) acc[t] = (f4){0, 0, 0, 0};
    float v[8];
    for (int j = 0; j < 8; j++) v[j] = 0.001f * (lane + j);
    if (ROLE == 0) body<F, NACC, TRANS>(acc, v, w, x, iters);
    else if (wv < NW / 2) body<0, NACC, false>(acc, v, w, x, iters);
    else {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int m = 0; m < 12 * F; m++) {
                if (TRANS && (m % F) == 0) v[m % 8] = __builtin_amdgcn_exp2f(v[m % 8]);
                else v[m % 8] = __builtin_fmaf(v[m % 8], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int t = 0; t < NACC; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    for (int j = 0; j < 8; j++) s += v[j];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

template <int F, int NACC, int NW, int ROLE, bool TRANS>
void run(const char *name)
{
    float *out;
    hipMalloc(&out, 256 * NW * 64 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<F, NACC, NW, ROLE, TRANS>), dim3(256), dim3(NW * 64), 0, 0, out, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<F, NACC, NW, ROLE, TRANS>), dim3(256), dim3(NW * 64), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_iter = ms * 1e6 / iters;
    const double mfma_per_simd = ROLE == 0 ? 12.0 * NW / 4 : 12.0 * NW / 8;
    printf("%-28s F=%d acc=%d waves/CU=%d%s: %.0f ns/iter -> %.2f ns per MFMA per SIMD (%.1f cyc @1.96GHz)\n", name, F, NACC, NW, TRANS ? " +exp" : "", ns_iter,
           ns_iter / mfma_per_simd, ns_iter / mfma_per_simd * 1.96);
    hipFree(out);
}

int main()
{
    run<0, 2, 8, 0, false>("same-wave");
    run<1, 2, 8, 0, false>("same-wave");
    run<2, 2, 8, 0, false>("same-wave");
    run<3, 2, 8, 0, false>("same-wave");
    run<4, 2, 8, 0, false>("same-wave");
    run<6, 2, 8, 0, false>("same-wave");
    run<3, 2, 8, 0, true>("same-wave");
    run<3, 1, 8, 0, false>("same-wave (1 acc chain)");
    run<3, 4, 8, 0, false>("same-wave (4 acc)");
    run<3, 2, 4, 0, false>("same-wave");
    run<0, 2, 8, 1, false>("cross-wave MFMA half only");
    run<2, 2, 8, 1, false>("cross-wave");
    run<3, 2, 8, 1, false>("cross-wave");
    run<6, 2, 8, 1, false>("cross-wave");
    run<6, 2, 8, 1, true>("cross-wave");
    return 0;
}
