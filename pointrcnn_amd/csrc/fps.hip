// Farthest point sampling + gather_point for gfx950 (MI355X).
//
// Replaces farthestpointsamplingKernel / gatherpointKernel of the reference
// (models/tf_ops/sampling/tf_sampling_g.cu:105-181).  Design (not a port):
//   * one workgroup per cloud (FPS is m-1 strictly serial iterations; a cross-CU
//     hand-off per iteration costs more than one CU's whole iteration);
//   * the cloud lives in VGPRs for the whole kernel (PPT points per lane), the running
//     min-distance array in VGPRs (n <= 16384) or in LDS (n <= 32768): no global or
//     LDS traffic in the distance loop;
//   * per iteration ONE barrier: wave argmax by DPP max + ballot (first lane = lowest
//     thread slot), then one 64-bit LDS atomic max per wave on (value, ~tie-break key)
//     into a rotating slot that every lane reads back after the barrier;
//   * distances of two points per packed fp32 instruction (v_pk_add/mul_f32: measured
//     faster than the scalar form on gfx950, and bit-identical per component).
// Bit-exactness contract: fp32 d=(dx*dx+dy*dy)+dz*dz without FMA
// contraction (file is built with -ffp-contract=off), temp=min(d,temp), winner =
// (max value, smallest k mod 512, smallest k).  Lane t of the workgroup owns reference
// thread slot s = t & 511 and the points k = s + 512*(h*PPT+q), h = t>>9, q ascending,
// so a lane's strict '>' scan is the reference thread's scan restricted to its half.
// Distances are non-negative floats (or the -1.0f "empty" sentinel), so all min / max /
// compare run on the int bit patterns (order-preserving, no NaN canonicalisation code).
#include "common.h"

#ifndef PRCNN_FPS_PROBE
#define PRCNN_FPS_PROBE 0      // diagnostic builds only: 1 = no coordinate load, 2 = no cross-wave step, 4 = no wave reduce
#endif

namespace prcnn {

typedef float f32x2 __attribute__((ext_vector_type(2)));

// value (non-negative float bits or the -1.0f sentinel) and tie-break key packed so that
// an unsigned 64-bit max picks (largest value, then smallest key)
__device__ __forceinline__ unsigned long long fps_pack(int vbits, unsigned key)
{
    const unsigned hi = vbits < 0 ? 0u : (unsigned)vbits + 1u;
    return ((unsigned long long)hi << 32) | (unsigned long long)(~key);
}

template <int PPT, int H, bool TD_LDS>
__global__ __launch_bounds__(512 * H) void fps_kernel(int n, int m,
                                                       const float *__restrict__ dataset,
                                                       int *__restrict__ idxs)
{
    constexpr int T = 512 * H;
    constexpr int P2 = (PPT + 1) / 2;      // points are processed in pairs (packed fp32 math)
    extern __shared__ __attribute__((aligned(16))) int smem[];
    // three rotating 64-bit slots for the cross-wave argmax (ds_max_u64), see below
    unsigned long long *slot = reinterpret_cast<unsigned long long *>(smem);
    int *tdl = smem + 8;                   // TD_LDS: [2*P2][T] running min distances

    const int tid = threadIdx.x;
    const int s = tid & 511;
    const int h = tid >> 9;
    const float *data = dataset + (size_t)blockIdx.x * n * 3;
    int *out = idxs + (size_t)blockIdx.x * m;

    f32x2 px[P2], py[P2], pz[P2];
    int td[TD_LDS ? 1 : 2 * P2];
#pragma unroll
    for (int q = 0; q < 2 * P2; ++q) {
        const int k = s + 512 * (h * PPT + q);
        const bool ok = q < PPT && k < n;
        px[q >> 1][q & 1] = ok ? data[k * 3 + 0] : 0.f;
        py[q >> 1][q & 1] = ok ? data[k * 3 + 1] : 0.f;
        pz[q >> 1][q & 1] = ok ? data[k * 3 + 2] : 0.f;
        const int init = ok ? f2i(1e38f) : f2i(-1.0f);   // -1: never beats best=-1
        if (TD_LDS) tdl[q * T + tid] = init; else td[q] = init;
    }
    if (tid < 3) slot[tid] = 0ull;
    if (tid == 0) out[0] = 0;
    int old = 0;
    __syncthreads();

    for (int j = 1; j < m; ++j) {
        const int so = __builtin_amdgcn_readfirstlane(old);
#if (PRCNN_FPS_PROBE & 1)
        const float x1 = so * 0.001f, y1 = so * 0.002f, z1 = so * 0.003f;
#else
        const float x1 = data[so * 3 + 0];
        const float y1 = data[so * 3 + 1];
        const float z1 = data[so * 3 + 2];
#endif
        int best = f2i(-1.0f);
        int bq = 0;
#pragma unroll
        for (int p = 0; p < P2; ++p) {
            // two points per packed instruction; each component is an IEEE fp32 op, so the
            // result equals the scalar expression (x2-x1)^2+(y2-y1)^2+(z2-z1)^2 bit for bit
            const f32x2 dx = px[p] - x1, dy = py[p] - y1, dz = pz[p] - z1;
            const f32x2 d = dx * dx + dy * dy + dz * dz;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = 2 * p + u;
                const int t0 = TD_LDS ? tdl[q * T + tid] : td[q];
                const int d2 = min(f2i(d[u]), t0);
                if (TD_LDS) { if (d2 != t0) tdl[q * T + tid] = d2; } else td[q] = d2;
                bq = d2 > best ? q : bq;
                best = max(best, d2);
            }
        }
        // reference: a slot with no strictly-better point reports besti = 0
        const int bk = best > f2i(-1.0f) ? s + 512 * (h * PPT + bq) : 0;
        // wave argmax: lanes are in slot order, so the first lane holding the max wins
#if (PRCNN_FPS_PROBE & 4)
        const int wmax = __builtin_amdgcn_readfirstlane(best);
        const int wk = __builtin_amdgcn_readfirstlane(bk);
#else
        const int wmax = wave_max_i32(best);
        const unsigned long long eq = __ballot(best == wmax);
        const int first = __builtin_ctzll(eq);
        const int wk = __builtin_amdgcn_readlane(bk, first);
#endif
        // cross-wave argmax: one LDS atomic max per wave on (value, ~(k mod 512, k / 512)).
        // Slot j%3 is used by iteration j; slot (j+2)%3 (last read in iteration j-1, next
        // written in iteration j+2) is cleared after this iteration's barrier.
#if (PRCNN_FPS_PROBE & 2)
        old = (wk + (wmax & 1)) % n;
#else
        const int cur = j % 3;
        if ((tid & 63) == 0)
            atomicMax(&slot[cur], fps_pack(wmax, (unsigned)(((wk & 511) << 8) | (wk >> 9))));
        __syncthreads();
        const unsigned long long win = slot[cur];
        if (tid == 0) slot[(j + 2) % 3] = 0ull;
        const unsigned k2 = ~(unsigned)win;
        old = (int)((k2 >> 8) + ((k2 & 255u) << 9));
#endif
        if (tid == 0) out[j] = old;
    }
}

// Clouds that do not fit one CU (n > 32768): same algorithm with the running
// min-distance array in the caller's `temp` buffer (the reference's scratch argument,
// tf_sampling.cpp:114-117) and the points re-read through L2 each iteration.
__global__ __launch_bounds__(1024) void fps_global_kernel(int b, int n, int m,
                                                           const float *__restrict__ dataset,
                                                           float *__restrict__ temp,
                                                           int *__restrict__ idxs)
{
    constexpr int T = 1024, NW = 16;
    __shared__ int rec[2 * NW * 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = tid & 511, h = tid >> 9;
    for (int i = blockIdx.x; i < b; i += gridDim.x) {
        const float *data = dataset + (size_t)i * n * 3;
        int *tdg = reinterpret_cast<int *>(temp) + (size_t)blockIdx.x * n;
        int *out = idxs + (size_t)i * m;
        for (int k = tid; k < n; k += T) tdg[k] = f2i(1e38f);
        int old = 0;
        if (tid == 0) out[0] = 0;
        // lane (s,h) walks k = s + 512*(2*r + h): ascending k inside one reference slot.
        for (int j = 1; j < m; ++j) {
            const int so = __builtin_amdgcn_readfirstlane(old);
            const float x1 = data[so * 3 + 0], y1 = data[so * 3 + 1], z1 = data[so * 3 + 2];
            int best = f2i(-1.0f), bk = 0;
            for (int k = s + 512 * h; k < n; k += T) {
                const float dx = data[k * 3 + 0] - x1, dy = data[k * 3 + 1] - y1,
                            dz = data[k * 3 + 2] - z1;
                const float d = dx * dx + dy * dy + dz * dz;
                const int t0 = tdg[k];
                const int d2 = min(f2i(d), t0);
                if (d2 != t0) tdg[k] = d2;
                bk = d2 > best ? k : bk;
                best = max(best, d2);
            }
            // lanes of one wave share h, so lane order = slot order; the two halves
            // (h=0,1) of a slot interleave in k, which the (k mod 512, k/512) key resolves.
            const int wmax = wave_max_i32(best);
            const unsigned long long eq = __ballot(best == wmax);
            const int first = __builtin_ctzll(eq);
            const int wk = __builtin_amdgcn_readlane(bk, first);
            const int par = j & 1;
            if (lane == 0) {
                rec[(par * NW + wave) * 2 + 0] = wmax;
                rec[(par * NW + wave) * 2 + 1] = wk;
            }
            __syncthreads();
            const int e = lane < NW ? lane : 0;
            const int ev = rec[(par * NW + e) * 2 + 0];
            const int ek = rec[(par * NW + e) * 2 + 1];
            const int mv = row16_allmax_i32(ev);
            const unsigned key = ev == mv ? (unsigned)(((ek & 511) << 8) | (ek >> 9)) : 0x7fffffffu;
            const unsigned mk = row16_allmin_u32(key);
            old = __builtin_amdgcn_readfirstlane((int)((mk >> 8) + ((mk & 255u) << 9)));
            if (tid == 0) out[j] = old;
        }
        __syncthreads();
    }
}

__global__ void gather_point_kernel(int n, int m, long long total, const float *__restrict__ inp,
                                    const int *__restrict__ idx, float *__restrict__ out)
{
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long row = e / 3;
        const int c = (int)(e - row * 3);
        const long long bi = row / m;
        const int a = idx[row];
        out[e] = inp[(bi * n + a) * 3 + c];
    }
}

template <int PPT, int H, bool TD_LDS>
static int launch_fps(int b, int n, int m, const float *inp, int *out, hipStream_t st)
{
    constexpr int T = 512 * H;
    size_t lds = sizeof(int) * (8 + (TD_LDS ? 2 * ((PPT + 1) / 2) * T : 0));
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)fps_kernel<PPT, H, TD_LDS>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return PRCNN_HIP_ERROR;
    }
    hipLaunchKernelGGL((fps_kernel<PPT, H, TD_LDS>), dim3(b), dim3(T), lds, st, n, m, inp, out);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

}  // namespace prcnn

extern "C" size_t prcnn_fps_temp_bytes(int b, int n)
{
    if (n <= 32768 || b <= 0) return 0;
    return sizeof(float) * (size_t)n * (size_t)(b < 32 ? b : 32);
}

extern "C" int prcnn_farthest_point_sample(int b, int n, int m, const float *inp, float *temp,
                                           int *out, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (m <= 0) return PRCNN_BAD_ARG;                 // tf_sampling.cpp:99 npoint > 0
    if (b < 0 || n <= 0) return PRCNN_BAD_SHAPE;      // tf_sampling.cpp:105
    if (b == 0) return PRCNN_OK;
    if (!inp || !out) return PRCNN_BAD_SHAPE;
    if (n > 512 * 256) return PRCNN_UNSUPPORTED;      // tie-break key packs k/512 in 8 bits
    hipStream_t st = to_stream(stream);
    if (n <= 512) return launch_fps<1, 1, false>(b, n, m, inp, out, st);
    if (n <= 1024) return launch_fps<1, 2, false>(b, n, m, inp, out, st);
    if (n <= 2048) return launch_fps<2, 2, false>(b, n, m, inp, out, st);
    if (n <= 4096) return launch_fps<4, 2, false>(b, n, m, inp, out, st);
    if (n <= 8192) return launch_fps<8, 2, false>(b, n, m, inp, out, st);
    if (n <= 16384) return launch_fps<16, 2, false>(b, n, m, inp, out, st);
    if (n <= 32768) return launch_fps<32, 2, true>(b, n, m, inp, out, st);
    if (!temp) return PRCNN_WORKSPACE;
    hipLaunchKernelGGL(fps_global_kernel, dim3(b < 32 ? b : 32), dim3(1024), 0, st, b, n, m, inp,
                       temp, out);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_gather_point(int b, int n, int m, const float *inp, const int *idx,
                                  float *out, prcnn_stream_t stream)
{
    if (b < 0 || n <= 0 || m < 0) return PRCNN_BAD_SHAPE;   // tf_sampling.cpp:131,135
    const long long total = (long long)b * m * 3;
    if (total == 0) return PRCNN_OK;
    if (!inp || !idx || !out) return PRCNN_BAD_SHAPE;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(prcnn::gather_point_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       to_stream(stream), n, m, total, inp, idx, out);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}
