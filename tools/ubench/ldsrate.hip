// Microbenchmark (experiment, round 6): ds_read_b128 service rate and round-trip time on gfx950, with the trunk's operand address pattern (lane group g
// reads 16 consecutive 16-byte slots of plane g) -- NR reads back to back per s_waitcnt lgkmcnt(0), 1 / 4 / 8 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NR>
__global__ __launch_bounds__(512) void k(float *out, int iters, int plane_slots, unsigned long long *cyc)
{
    __shared__ __attribute__((aligned(16))) float L[32768];     // 128 KB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) L[i] = 0.001f * (i & 63);
    __syncthreads();
    const int g = lane >> 4, c16 = lane & 15;
    unsigned a0 = ((g * plane_slots + c16 + wv * 16) * 16) & 0x1fff0;
    f4 r[NR];
    f4 s = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < NR; q++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[q]) : "v"(a0), "n"(q * 256 * 4));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < NR; q++) asm volatile("" ::"v"(r[q]));
        s += r[0];
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (blockIdx.x == 5 && threadIdx.x == 0) *cyc = t1 - t0;
}

template <int NR>
void run(float *out, unsigned long long *cyc, int threads, int plane)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NR>), dim3(256), dim3(threads), 0, 0, out, 50, plane, cyc);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NR>), dim3(256), dim3(threads), 0, 0, out, iters, plane, cyc);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double per_iter_cyc = (double)c / iters, ns = ms * 1e6 / iters;
    printf("%2d reads per wait, %d waves/CU, plane pitch %4d slots: %7.1f counter ticks per round (%5.1f per read per wave), %7.1f ns per round -> %5.2f ns per ds_read_b128 per CU\n",
           NR, threads / 64, plane, per_iter_cyc, per_iter_cyc / NR, ns, ns / (NR * (threads / 64)));
}

int main()
{
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&cyc, 8);
    for (int plane : {176, 184}) {
        for (int threads : {64, 256, 512}) {
            run<1>(out, cyc, threads, plane);
            run<2>(out, cyc, threads, plane);
            run<4>(out, cyc, threads, plane);
            run<8>(out, cyc, threads, plane);
            run<16>(out, cyc, threads, plane);
        }
    }
    return 0;
}
