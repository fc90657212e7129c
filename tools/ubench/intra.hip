// Microbenchmark (experiment): how many independent VALU / DS instructions hide behind v_mfma_f32_16x16x32_f16 when they are
// interleaved IN THE SAME WAVE's instruction stream ({MFMA, NF fillers} repeated, 4 independent accumulators), with one and
// with two such waves per SIMD.  Companion of coexec2.hip (fillers in the OTHER wave of the SIMD: no overlap).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

enum { OP_FMA, OP_EXP, OP_CVT, OP_MIX, OP_DSW, OP_DSR, OP_N };
static const char *OP_NAME[OP_N] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "ds_write_b64", "ds_read_b128"};

template <int OP>
__device__ __forceinline__ void fill(float &v, float &p, int lane, int j)
{
    if (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(p));
    if (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
    if (OP == OP_CVT) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(v) : "v"(p));
    if (OP == OP_MIX) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(v) : "v"(p));
    if (OP == OP_DSW) asm volatile("ds_write_b64 %0, %1" ::"v"(lane * 8 + (j & 7) * 512), "v"(*reinterpret_cast<double *>(&p)) : "memory");
    if (OP == OP_DSR) {
        f4 t;
        asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lane * 16 + (j & 7) * 1024) : "memory");
    }
}

template <int OP, int NF, bool MFMA>
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
    __shared__ __attribute__((aligned(16))) _Float16 lds[16384];
    const int lane = threadIdx.x & 63;
    h8 w, x;
    for (int j = 0; j < 8; j++) { w[j] = (_Float16)(0.01f * (j + lane % 3)); x[j] = (_Float16)(0.02f * (j + lane % 5)); }
    float v[8], p[2] = {1.0001f, 1.0002f};
    for (int j = 0; j < 8; j++) v[j] = 0.001f * (lane + j);
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = (_Float16)0.0f;
    __syncthreads();
    f4 acc[4] = {};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 16; m++) {
            if (MFMA) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(w), "v"(x));
#pragma unroll
            for (int f = 0; f < NF; f++) fill<OP>(v[(m * NF + f) & 7], p[0], lane, m * NF + f);
        }
        if (OP == OP_DSR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    for (int j = 0; j < 8; j++) s += v[j];
    out[blockIdx.x * 512 + threadIdx.x] = s + (float)lds[threadIdx.x];
}

template <int OP, int NF, bool MFMA>
float time_it(float *out, int threads)
{
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, NF, MFMA>), dim3(256), dim3(threads), 0, 0, out, 50);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, NF, MFMA>), dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / iters / 16;                                                // ns per {MFMA + NF fillers}
}

template <int OP, int NF>
void run(float *out)
{
    for (int threads = 256; threads <= 512; threads += 256) {
        const float tb = time_it<OP, NF, true>(out, threads), tf = time_it<OP, NF, false>(out, threads);
        printf("%-18s x%d per MFMA, %d wave(s)/SIMD: mfma+fillers %6.2f ns per group per wave-slot, fillers alone %6.2f\n", OP_NAME[OP], NF, threads / 256, tb, tf);
    }
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * 4);
    run<OP_FMA, 0>(out);
    run<OP_FMA, 1>(out); run<OP_FMA, 2>(out); run<OP_FMA, 3>(out); run<OP_FMA, 4>(out); run<OP_FMA, 6>(out);
    run<OP_EXP, 1>(out); run<OP_EXP, 2>(out); run<OP_EXP, 3>(out);
    run<OP_CVT, 2>(out); run<OP_MIX, 2>(out); run<OP_MIX, 3>(out);
    run<OP_DSW, 1>(out); run<OP_DSW, 2>(out); run<OP_DSR, 1>(out); run<OP_DSR, 2>(out);
    hipFree(out);
    return 0;
}
