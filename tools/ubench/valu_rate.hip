// Microbenchmark (experiment): issue rate of the integer VALU instructions the alignment kernels are made of -- 32-bit
// (v_sub_u32, v_max_i32, v_cndmask) against packed 16-bit (v_pk_sub_i16, v_pk_max_i16, v_pk_ashrrev_i16, v_pk_mad_i16, v_bfi_b32) --
// as ns per wave-instruction per SIMD, with W waves per SIMD and 8 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
enum { O_SUB32, O_MAX32, O_PKSUB, O_PKMAX, O_PKASHR, O_PKMAD, O_BFI, O_XOR, O_PKADD, O_N };
static const char *NAME[O_N] = {"v_sub_u32", "v_max_i32", "v_pk_sub_i16", "v_pk_max_i16", "v_pk_ashrrev_i16", "v_pk_mad_i16", "v_bfi_b32", "v_xor_b32", "v_pk_add_i16"};
template <int OP> __device__ __forceinline__ void op(uint32_t &v, uint32_t p)
{
    if (OP == O_SUB32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v) : "v"(p));
    if (OP == O_MAX32) asm volatile("v_max_i32 %0, %0, %1" : "+v"(v) : "v"(p));
    if (OP == O_PKSUB) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(v) : "v"(p));
    if (OP == O_PKMAX) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(v) : "v"(p));
    if (OP == O_PKASHR) asm volatile("v_pk_ashrrev_i16 %0, 1, %0 op_sel_hi:[0,1]" : "+v"(v));
    if (OP == O_PKMAD) asm volatile("v_pk_mad_i16 %0, %0, %1, %1" : "+v"(v) : "v"(p));
    if (OP == O_BFI) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(v) : "v"(p));
    if (OP == O_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v) : "v"(p));
    if (OP == O_PKADD) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(v) : "v"(p));
}
template <int OP> __global__ void k(uint32_t *out, int iters)
{
    uint32_t v[8], p = threadIdx.x * 2654435761u + 12345;
    for (int j = 0; j < 8; j++) v[j] = threadIdx.x + j * 77;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < 64; m++) op<OP>(v[m & 7], p);
    }
    uint32_t s = 0;
    for (int j = 0; j < 8; j++) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int OP> void run(uint32_t *out, int wps)
{
    const int iters = 4000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const dim3 grid(256), block(256 * wps);                   // one workgroup per CU, wps waves per SIMD
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double n_per_simd = (double)iters * 64 * wps;
    printf("%-18s %d waves/SIMD: %.3f ns per wave-instruction per SIMD\n", NAME[OP], wps, ms * 1e6 / n_per_simd);
}
int main()
{
    uint32_t *out;
    hipMalloc(&out, 256 * 1024 * 4);
    for (int wps : {1, 2, 4}) {
        run<O_SUB32>(out, wps); run<O_MAX32>(out, wps); run<O_XOR>(out, wps); run<O_BFI>(out, wps); run<O_PKSUB>(out, wps); run<O_PKADD>(out, wps);
        run<O_PKMAX>(out, wps); run<O_PKASHR>(out, wps); run<O_PKMAD>(out, wps);
    }
    return 0;
}
