// Diagnostic only (bench.py PRCNN_BENCH_PROBE_NO_FPS=7x): a kernel with the small-footprint sampler's
// shape (one 256-thread workgroup per cloud) that stays resident for a given time and does
//   mode 0: nothing (s_sleep),  mode 1: a dependent chain of global loads,  mode 2: VALU work,
//   mode 3: mode 1 plus one 8-byte global store per lane and iteration (sink must hold 128 + blocks * 512 long longs),
//   mode 4: the sampler's per-iteration LDS atomic + workgroup barrier + LDS read, nothing else.
// Bounded: every wave leaves after `ticks` of the 100 MHz clock or max_it iterations, whichever is first.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void spin_kernel(long long ticks, int max_it, int mode, const int *__restrict__ ring,
                                                   int ring_n, int *__restrict__ sink)
{
    const long long t0 = wall_clock64();
    int pos = (blockIdx.x * 256 + threadIdx.x) % ring_n;
    float acc = threadIdx.x;
    for (int it = 0; it < max_it && wall_clock64() - t0 < ticks; ++it) {
        if (mode == 0) {
            __builtin_amdgcn_s_sleep(32);
        } else if (mode == 1) {
            pos = ring[pos] % ring_n;                 // dependent load: one L2 round trip per iteration
            pos = pos < 0 ? 0 : pos;
        } else if (mode == 4) {
            // the sampler's synchronisation skeleton: one LDS atomic per wave, a workgroup barrier, an LDS read, ~1 us apart
            __shared__ unsigned long long slot[3];
            if ((threadIdx.x & 63) == 0) atomicMax(&slot[it % 3], (unsigned long long)(it + threadIdx.x));
            __syncthreads();
            pos = (int)(slot[it % 3] & 1023);
            if (threadIdx.x == 0) slot[(it + 2) % 3] = 0ull;
            __builtin_amdgcn_s_sleep(24);
        } else if (mode == 3) {
            pos = ring[pos] % ring_n;                 // the load chain of mode 1 + one 8-byte store per lane (dirty L2 lines)
            pos = pos < 0 ? 0 : pos;
            reinterpret_cast<long long *>(sink)[64 + (size_t)(blockIdx.x * 256 + threadIdx.x) * 2 + (it & 1)] = pos;
        } else {
#pragma unroll
            for (int k = 0; k < 64; ++k) acc = acc * 1.0001f + 0.5f;
        }
    }
    if (acc == -1.f || pos == -7) sink[0] = 1;        // never true: keeps the work alive
}

extern "C" int spin_launch(int blocks, double ms, int mode, const int *ring, int ring_n, int *sink, void *stream)
{
    if (blocks <= 0 || ring_n <= 0 || !ring || !sink) return -1;
    const long long ticks = (long long)(ms * 1e5);    // 100 MHz
    const int max_it = (int)(ms * 20000.0) + 1;
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ticks, max_it, mode, ring, ring_n, sink);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
