// Probe kernels for co-residency experiments (tools/exp_probe.py): each workgroup records the wall clock at its start and spins
// for `spin_ns`; R = VGPRs kept live per lane, LDS bytes by dynamic shared memory.  extern "C" launcher for ctypes.
#include <hip/hip_runtime.h>
#include <cstdint>
template <int R>
__global__ __launch_bounds__(256) void k_probe(uint64_t *start, uint64_t *stop, float *sink, int spin_ns)
{
    extern __shared__ float dyn[];
    float v[R];
#pragma unroll
    for (int i = 0; i < R; i++) v[i] = threadIdx.x * 0.5f + i;
    const uint64_t t0 = wall_clock64();                     // 100 MHz
    if (threadIdx.x == 0) start[blockIdx.x] = t0;
    while ((wall_clock64() - t0) * 10 < (uint64_t)spin_ns) {
#pragma unroll
        for (int i = 0; i < R; i++) v[i] = v[i] * 1.0001f + 0.5f;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < R; i++) s += v[i];
    if (threadIdx.x == 0) stop[blockIdx.x] = wall_clock64();
    if (s == 12345.678f) sink[0] = s + dyn[0];
}
extern "C" int probe_launch(void *stream, int regs, int lds_bytes, int blocks, int threads, uint64_t *start, uint64_t *stop, float *sink, int spin_ns)
{
#define L(RR) hipLaunchKernelGGL(k_probe<RR>, dim3(blocks), dim3(threads), lds_bytes, (hipStream_t)stream, start, stop, sink, spin_ns)
    switch (regs) { case 8: L(8); break; case 11: L(11); break; case 14: L(14); break; case 17: L(17); break; case 20: L(20); break; case 23: L(23); break; case 26: L(26); break; case 30: L(30); break; default: L(100); }
    return (int)hipGetLastError();
}
extern "C" uint64_t probe_now_kernelless() { return 0; }
