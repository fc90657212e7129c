// Microbenchmark (experiment, not product): sustained rate of {ds_read_b128 x R -> v_mfma_f32_16x16x32_f16 x M} loops on gfx950.
// build: hipcc -O3 --offload-arch=gfx950 mfma_lds.hip -o mfma_lds ; run: ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only (operands in registers).  MODE 1: operands read from LDS one iteration ahead.  MODE 2: LDS reads only.
template <int MODE, int NT, int NW>
__global__ __launch_bounds__(NW * 64) void k(float *out, int iters, int stride_slots)
{
    __shared__ __attribute__((aligned(16))) _Float16 L[2][4096 * 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += NW * 64)
        for (int j = 0; j < 8; j++) { L[0][i * 8 + j] = (_Float16)(0.001f * (i % 7)); L[1][i * 8 + j] = (_Float16)(0.002f * (i % 5)); }
    __syncthreads();
    h8 w[6];
    for (int q = 0; q < 6; q++)
        for (int j = 0; j < 8; j++) w[q][j] = (_Float16)(0.01f * (q + j + lane % 3));
    f4 acc[NT];
    for (int t = 0; t < NT; t++) acc[t] = (f4){0, 0, 0, 0};
    int base = ((lane & 15) * stride_slots + (lane >> 4) * 221 + wv * 16) * 8;
    h8 a[2][NT], b[2][NT];
    auto load = [&](int it, int slot) {
        const int off = ((it * 3) & 1023) * 8;
        for (int t = 0; t < NT; t++) {
            a[slot][t] = *reinterpret_cast<const h8 *>(&L[0][(base + off + t * 16 * 8) & (4096 * 8 - 8)]);
            b[slot][t] = *reinterpret_cast<const h8 *>(&L[1][(base + off + t * 16 * 8) & (4096 * 8 - 8)]);
        }
    };
    if (MODE != 0) load(0, 0);
    else for (int t = 0; t < NT; t++) { a[0][t] = w[0]; b[0][t] = w[1]; a[1][t] = w[2]; b[1][t] = w[3]; }
    auto step = [&](int it, const int cur) {
        if (MODE != 0) load(it + 1, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 2) {
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[4], a[cur][t], acc[t], 0, 0, 0);
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[4], b[cur][t], acc[t], 0, 0, 0);
            for (int t = 0; t < NT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[5], a[cur][t], acc[t], 0, 0, 0);
        } else {
            for (int t = 0; t < NT; t++) { acc[t][0] += (float)a[cur][t][0]; acc[t][1] += (float)b[cur][t][1]; }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int it = 0; it < iters; it += 2) { step(it, 0); step(it + 1, 1); }
    float s = 0;
    for (int t = 0; t < NT; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * NW * 64 + threadIdx.x] = s;
}

template <int MODE, int NT, int NW>
void run(const char *name, int stride)
{
    float *out;
    hipMalloc(&out, 256 * NW * 64 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NT, NW>), dim3(256), dim3(NW * 64), 0, 0, out, 100, stride);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NT, NW>), dim3(256), dim3(NW * 64), 0, 0, out, iters, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / iters;
    printf("%-44s NT=%d waves/CU=%d stride=%d: %.1f cycles/iter (%d MFMA + %d ds_read_b128 per wave-iter) -> %.1f cyc/MFMA/SIMD\n", name, NT, NW, stride, cyc,
           MODE == 2 ? 0 : 3 * NT, MODE == 0 ? 0 : 2 * NT, MODE == 2 ? 0.0 : cyc / (3 * NT * (NW / 4.0)));
    hipFree(out);
}

int main()
{
    run<0, 3, 4>("MFMA only", 1);
    run<0, 3, 8>("MFMA only", 1);
    run<0, 2, 8>("MFMA only", 1);
    run<0, 1, 8>("MFMA only", 1);
    run<2, 3, 8>("LDS reads only", 1);
    run<2, 3, 8>("LDS reads only (stride 2 slots)", 2);
    run<1, 3, 4>("LDS -> MFMA", 1);
    run<1, 3, 8>("LDS -> MFMA", 1);
    run<1, 2, 8>("LDS -> MFMA", 1);
    run<1, 1, 8>("LDS -> MFMA", 1);
    run<1, 3, 8>("LDS -> MFMA (stride 2 slots)", 2);
    return 0;
}
