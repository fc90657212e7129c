// Microbenchmark (experiment, round 6): how much of the trunk's REAL epilogue instruction mix (v_mul, v_exp, v_med3, 2 x v_fma, v_fma_mix, v_cvt_pk per
// activation) hides behind v_mfma_f32_16x16x32_f16 when it is interleaved IN THE SAME WAVE's stream, as a function of the number of accumulators the
// MFMAs rotate over (the trunk's loops have 2-5) and of the waves per SIMD.  {MFMA, NF vector instructions} repeated; per MFMA-group time in ns and in
// shader cycles (s_memtime).  Companion of intra.hip (single opcodes, 4 accumulators) and epi.hip (the epilogue in the OTHER wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int J>
__device__ __forceinline__ void filler(float (&v)[8], float c, uint32_t &pk)
{
    constexpr int op = J % 7, r = (J / 7) % 4;
    if (op == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[r]) : "v"(v[r + 4]), "v"(c));
    if (op == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[r]));
    if (op == 2) asm volatile("v_med3_f32 %0, %1, 0, %2" : "=v"(v[r + 4]) : "v"(v[r + 4]), "v"(c));
    if (op == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(c));
    if (op == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v[r]) : "v"(v[r + 4]), "v"(c));
    if (op == 5) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(pk) : "v"(v[r]));
    if (op == 6) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(v[r + 4]) : "v"(pk), "v"(v[r]));
}
template <int NF, int BASE, int I = 0>
__device__ __forceinline__ void fillers(float (&v)[8], float c, uint32_t &pk)
{
    if constexpr (I < NF) { filler<BASE + I>(v, c, pk); fillers<NF, BASE, I + 1>(v, c, pk); }
}

template <int NACC, int NF, bool MFMA, int M = 0>
__device__ __forceinline__ void body(f4 (&acc)[4], const h8 &w, const h8 &x, float (&v)[8], float c, uint32_t &pk)
{
    if constexpr (M < 28) {
        if (MFMA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[M % NACC]) : "v"(w), "v"(x));
        fillers<NF, M * NF>(v, c, pk);
        body<NACC, NF, MFMA, M + 1>(acc, w, x, v, c, pk);
    }
}

template <int NACC, int NF, bool MFMA>
__global__ __launch_bounds__(512) void k(float *out, int iters, unsigned long long *cyc)
{
    const int lane = threadIdx.x & 63;
    h8 w, x;
    for (int j = 0; j < 8; j++) { w[j] = (_Float16)(0.01f * (j + lane % 3)); x[j] = (_Float16)(0.02f * (j + lane % 5)); }
    float v[8];
    for (int j = 0; j < 8; j++) v[j] = -0.001f * (lane + j);
    uint32_t pk = 0;
    f4 acc[4] = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) body<NACC, NF, MFMA>(acc, w, x, v, 0.999f, pk);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)pk;
    for (int j = 0; j < 8; j++) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (blockIdx.x == 7 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int NF, bool MFMA>
void time_it(float *out, unsigned long long *cyc, int threads, float &ns, float &cy)
{
    const int iters = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, NF, MFMA>), dim3(256), dim3(threads), 0, 0, out, 20, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, NF, MFMA>), dim3(256), dim3(threads), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    ns = ms * 1e6f / iters / 28;
    cy = (float)c / iters / 28;
}

template <int NACC, int NF>
void run(float *out, unsigned long long *cyc)
{
    for (int threads = 256; threads <= 512; threads += 256) {
        float a, ac, b = 0, bc = 0;
        time_it<NACC, NF, true>(out, cyc, threads, a, ac);
        if (NF) time_it<NACC, NF, false>(out, cyc, threads, b, bc);
        printf("%d accumulators, %d epilogue instructions per MFMA, %d wave(s)/SIMD: %6.2f ns = %5.1f cycles per MFMA group per wave; the vector instructions alone %6.2f ns = %5.1f cycles\n",
               NACC, NF, threads / 256, a, ac, b, bc);
    }
}

int main()
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    run<4, 0>(out, cyc); run<2, 0>(out, cyc); run<1, 0>(out, cyc);
    run<4, 1>(out, cyc); run<4, 2>(out, cyc); run<4, 3>(out, cyc); run<4, 4>(out, cyc); run<4, 6>(out, cyc);
    run<2, 1>(out, cyc); run<2, 2>(out, cyc); run<2, 3>(out, cyc); run<2, 4>(out, cyc);
    run<1, 2>(out, cyc); run<1, 3>(out, cyc);
    hipFree(out);
    return 0;
}
