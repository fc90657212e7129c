// Microbenchmark (experiment, round 6): the trunk's epilogue (SELU + fp16 hi / lo split + ds_write_b64 per accumulator of 4 values) ALONE: cycles per
// activation as a function of how many accumulators are processed in lockstep (instruction-level parallelism of the dependent chains
// mul -> exp -> fma -> fma -> cvt_pk -> fma_mix -> cvt_pk -> ds_write), one and two waves per SIMD.  BATCH = 1 is the product's order (one accumulator
// after the other).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define SELU_LA 1.7580993408473766f
struct h_epi { float c1, c2, c3; };
__device__ __forceinline__ float exp2_clamp01(float x) { float r; asm("v_exp_f32_e64 %0, %1 clamp" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ float sub_h_lo(float v, uint32_t hpk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v)); return r; }
__device__ __forceinline__ float sub_h_hi(float v, uint32_t hpk) { float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v)); return r; }

template <int NB>
__device__ __forceinline__ void epi_batch(const f4 (&acc)[NB], const h_epi &k, _Float16 *hp, _Float16 *lp)
{
    float a[NB * 4], e[NB * 4], pos[NB * 4], s[NB * 4];
#pragma unroll
    for (int i = 0; i < NB * 4; i++) a[i] = acc[i / 4][i % 4];
#pragma unroll
    for (int i = 0; i < NB * 4; i++) e[i] = a[i] * k.c1;
#pragma unroll
    for (int i = 0; i < NB * 4; i++) e[i] = exp2_clamp01(e[i]);
#pragma unroll
    for (int i = 0; i < NB * 4; i++) pos[i] = __builtin_amdgcn_fmed3f(a[i], 0.0f, k.c3);
#pragma unroll
    for (int i = 0; i < NB * 4; i++) e[i] = fmaf(e[i], SELU_LA, -SELU_LA);
#pragma unroll
    for (int i = 0; i < NB * 4; i++) s[i] = fmaf(pos[i], k.c2, e[i]);
    uint32_t u[NB * 2], l[NB * 2];
#pragma unroll
    for (int i = 0; i < NB * 2; i++) u[i] = __builtin_bit_cast(uint32_t, (h2)__builtin_convertvector((f2){s[2 * i], s[2 * i + 1]}, h2));
    float d[NB * 4];
#pragma unroll
    for (int i = 0; i < NB * 2; i++) { d[2 * i] = sub_h_lo(s[2 * i], u[i]); d[2 * i + 1] = sub_h_hi(s[2 * i + 1], u[i]); }
#pragma unroll
    for (int i = 0; i < NB * 2; i++) l[i] = __builtin_bit_cast(uint32_t, (h2)__builtin_convertvector((f2){d[2 * i], d[2 * i + 1]}, h2));
#pragma unroll
    for (int q = 0; q < NB; q++) {
        *reinterpret_cast<uint2 *>(hp + q * 256) = make_uint2(u[2 * q], u[2 * q + 1]);
        *reinterpret_cast<uint2 *>(lp + q * 256) = make_uint2(l[2 * q], l[2 * q + 1]);
    }
}

template <int NA, int NB>
__global__ __launch_bounds__(512) void k(float *out, int iters, float c1, float c2, float c3, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[32768];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (_Float16)0.0f;
    __syncthreads();
    const h_epi e = {c1, c2, c3};
    f4 a[NA];
    for (int q = 0; q < NA; q++) a[q] = (f4){0.1f * lane - 3.0f + q, 0.2f * lane - 5.0f, 0.01f * q, -0.3f * lane};
    _Float16 *hp = lds + (wv & 3) * 4096 + lane * 4, *lp = hp + 16384;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < NA; q++) asm volatile("" : "+v"(a[q]));
#pragma unroll
        for (int q = 0; q < NA; q += NB) {
            f4 b[NB];
#pragma unroll
            for (int j = 0; j < NB; j++) b[j] = a[q + j];
            epi_batch<NB>(b, e, hp + q * 256, lp + q * 256);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = a[0][0] + (float)lds[threadIdx.x];
    if (blockIdx.x == 5 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int NA, int NB>
void run(float *out, unsigned long long *cyc)
{
    for (int threads = 256; threads <= 512; threads += 256) {
        const int iters = 2000;
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL((k<NA, NB>), dim3(256), dim3(threads), 0, 0, out, 50, 0.01f, 0.02f, 1000.0f, cyc);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NA, NB>), dim3(256), dim3(threads), 0, 0, out, iters, 0.01f, 0.02f, 1000.0f, cyc);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c;
        (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("%2d accumulators, %d in lockstep, %d wave(s)/SIMD: %6.1f ticks = %6.2f ns per activation per wave\n", NA, NB, threads / 256, (double)c / iters / (4 * NA), ms * 1e6 / iters / (4 * NA));
    }
}

int main()
{
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&cyc, 8);
    run<12, 1>(out, cyc); run<12, 2>(out, cyc); run<12, 3>(out, cyc); run<12, 4>(out, cyc); run<12, 6>(out, cyc);
    return 0;
}
