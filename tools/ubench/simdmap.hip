// Microbenchmark (experiment): which waves of a 512-thread workgroup share a SIMD on gfx950?  The waves named by a bit mask run a chain of
// v_mfma_f32_16x16x32_f16; two waves on one SIMD take twice the time of one.  VG: keep ~230 VGPRs alive (the trunk kernel's allocation).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int BIG>
__global__ __launch_bounds__(512) void k(float *out, int iters, unsigned mask)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    h8 w, x;
    for (int j = 0; j < 8; j++) { w[j] = (_Float16)(0.01f * (j + lane % 3)); x[j] = (_Float16)(0.02f * (j + lane % 5)); }
    f4 acc[4] = {};
    float keep[BIG ? 200 : 1];
    for (int j = 0; j < (BIG ? 200 : 1); j++) keep[j] = out[(lane + j) & 1023];
    if (mask >> wv & 1) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int m = 0; m < 16; m++) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(w), "v"(x));
        }
    }
    float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    for (int j = 0; j < (BIG ? 200 : 1); j++) asm volatile("" : "+v"(keep[j]));
    for (int j = 0; j < (BIG ? 200 : 1); j++) s += keep[j];
    if (threadIdx.x == 0) lds[0] = s;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int BIG>
void run(float *out, unsigned mask, size_t ldsb)
{
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void *)k<BIG>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    hipLaunchKernelGGL((k<BIG>), dim3(256), dim3(512), ldsb, 0, out, 50, mask);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<BIG>), dim3(256), dim3(512), ldsb, 0, out, iters, mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("VGPR-heavy %d, LDS %6zu B, waves 0x%02x: %.2f ns per MFMA of a wave\n", BIG, ldsb, mask, ms * 1e6f / iters / 16);
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    hipMemset(out, 0, 256 * 512 * 4);
    const unsigned masks[] = {0x01, 0x03, 0x05, 0x11, 0x09, 0x21, 0x0f, 0x33, 0x55, 0xf0, 0xff};
    for (unsigned m : masks) run<0>(out, m, 1024);
    for (unsigned m : masks) run<1>(out, m, 148736);
    hipFree(out);
    return 0;
}
