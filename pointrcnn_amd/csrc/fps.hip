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
#include "fps_sort.h"
#include <mutex>

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

// Clouds that do not fit one CU (n > 16384): G workgroups per cloud, each holding a
// 16384-point slice in registers exactly as above.  Per iteration every workgroup
// publishes its best candidate as ONE 8-byte data-tagged granule
//   [63:32] value bits + 1   [31:15] ~(k mod 512, k / 512)   [14:0] iteration tag
// with a single agent-scope (sc1, write-through) store and polls the G granules of its
// cloud with agent-scope loads until every tag equals the iteration: no counters, no
// resets, no fences (an 8-byte granule written by one store is never observed torn; the
// tag makes a stale value unmistakable).  The granule array (part of `temp`) is zeroed by
// a hipMemsetAsync ahead of the launch; tag 0 is never used before the slot has been
// overwritten.  All G workgroups of a cloud must be resident together: the host keeps the
// grid <= 256 workgroups (one 1024-thread workgroup per CU) and every spin is bounded
// (on time-out the error word in `temp` is set and the kernel leaves).
constexpr int COOP_PPT = 16;
constexpr int COOP_SLICE = 1024 * COOP_PPT;

__global__ __launch_bounds__(1024) void fps_coop_kernel(int n, int m, int G, int cloud0,
                                                        const float *__restrict__ dataset,
                                                        unsigned long long *__restrict__ gran,
                                                        int *__restrict__ err,
                                                        int *__restrict__ idxs, int spin_limit)
{
    constexpr int PPT = COOP_PPT, P2 = PPT / 2;
    __shared__ unsigned long long slot[3];
    __shared__ unsigned long long bcast[2];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int s = tid & 511, h = tid >> 9;
    const int cloud = cloud0 + blockIdx.x / G;
    const int g = blockIdx.x % G;
    const int base = g * COOP_SLICE;
    const float *data = dataset + (size_t)cloud * n * 3;
    int *out = idxs + (size_t)cloud * m;
    // two granule sets by iteration parity: a fast workgroup may publish iteration j+1 while a
    // slow one still reads iteration j (it cannot get two ahead: j+1 needs everyone's j+1)
    unsigned long long *cgran = gran + (size_t)cloud * 2 * G;

    f32x2 px[P2], py[P2], pz[P2];
    int td[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int k = base + s + 512 * (h * PPT + q);
        const bool ok = k < n;
        px[q >> 1][q & 1] = ok ? data[k * 3 + 0] : 0.f;
        py[q >> 1][q & 1] = ok ? data[k * 3 + 1] : 0.f;
        pz[q >> 1][q & 1] = ok ? data[k * 3 + 2] : 0.f;
        td[q] = ok ? f2i(1e38f) : f2i(-1.0f);
    }
    if (tid < 3) slot[tid] = 0ull;
    if (tid == 0 && g == 0) out[0] = 0;
    int old = 0;
    __syncthreads();

    for (int j = 1; j < m; ++j) {
        const int so = __builtin_amdgcn_readfirstlane(old);
        const float x1 = data[so * 3 + 0], y1 = data[so * 3 + 1], z1 = data[so * 3 + 2];
        int best = f2i(-1.0f), bq = 0;
#pragma unroll
        for (int p = 0; p < P2; ++p) {
            const f32x2 dx = px[p] - x1, dy = py[p] - y1, dz = pz[p] - z1;
            const f32x2 d = dx * dx + dy * dy + dz * dz;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = 2 * p + u;
                const int d2 = min(f2i(d[u]), td[q]);
                td[q] = d2;
                bq = d2 > best ? q : bq;
                best = max(best, d2);
            }
        }
        const int bk = best > f2i(-1.0f) ? base + s + 512 * (h * PPT + bq) : 0;
        const int wmax = wave_max_i32(best);
        const unsigned long long eq = __ballot(best == wmax);
        const int wk = __builtin_amdgcn_readlane(bk, __builtin_ctzll(eq));
        const int cur = j % 3;
        if (lane == 0)
            atomicMax(&slot[cur], fps_pack(wmax, (unsigned)(((wk & 511) << 8) | (wk >> 9))));
        __syncthreads();
        if (wave == 0) {
            const unsigned long long mine = slot[cur];                  // (value+1 | 0, ~key32)
            const unsigned tag = (unsigned)j & 0x7fffu;
            if (lane == 0) {
                const unsigned long long v = (mine & 0xffffffff00000000ull) |
                                             ((mine & 0x1ffffull) << 15) | tag;
                __hip_atomic_store(&cgran[(j & 1) * G + g], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned long long v = 0ull;
            int spins = 0;
            bool timeout = false;
            for (;;) {
                if (lane < G)
                    v = __hip_atomic_load(&cgran[(j & 1) * G + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool ready = lane >= G || ((unsigned)v & 0x7fffu) == tag;
                if (__all(ready)) break;
                if (++spins > spin_limit) { timeout = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (timeout) {
                if (lane == 0) { atomicExch(err, 1); bcast[j & 1] = ~0ull; }
            } else {
                // max over the G granules of (value, ~key): drop the tag, then a 64-bit butterfly
                unsigned long long w = lane < G ? (v >> 15) : 0ull;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const unsigned long long o = __shfl_xor(w, off);
                    w = o > w ? o : w;
                }
                if (lane == 0) bcast[j & 1] = w;
            }
        }
        __syncthreads();
        const unsigned long long win = bcast[j & 1];
        if (win == ~0ull) {                                              // time-out: leave (uniform)
            if (g == 0)                                                  // no sample from here on: -1, never garbage
                for (int q = j + tid; q < m; q += 1024) out[q] = -1;
            return;
        }
        if (tid == 0) slot[(j + 2) % 3] = 0ull;
        const unsigned k2 = ~(unsigned)win & 0x1ffffu;
        old = (int)((k2 >> 8) + ((k2 & 255u) << 9));
        if (tid == 0 && g == 0) out[j] = old;
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


// ---- pruned single-workgroup kernel (4096 < n <= 16384, long runs) ---------------------------
// The plain kernel above is VALU-throughput bound: every iteration touches all n points.  Here
// the cloud is first counting-sorted by grid cell inside the workgroup (fps_sort.h: 14-bit anisotropic
// Morton key, histogram in LDS) and cut into buckets of 128 consecutive sorted points; bucket
// g belongs to wave g%16 as its point pair g/16 (2 points per lane), so neighbouring buckets
// sit on different waves.  Each bucket keeps its bounding box, the maximum T of its running
// min-distances and the tie-break key K of that maximum.  An iteration updates only buckets
// whose box-to-sample lower bound is below T; the bound uses the same unfused fp32 operations
// as the distance and every one of them is monotone under rounding, so bound <= computed
// distance of every point in the box: a skipped bucket provably has no point with d < td.
// The cached (T,K) of a bucket is recomputed only when a holder of T was lowered.  Winner =
// (max value, smallest k mod 512, smallest k) on ORIGINAL indices, carried per point as a key,
// so the result is independent of the sorted layout and bit-identical to fps_kernel.

__device__ __forceinline__ int wave_min_i32_(int v) { return ~wave_max_i32(~v); }

// COOP: the workgroup holds slice g of a cloud of more than 16384 points (see fps_coop_kernel
// below in the file for the exchange protocol: one tagged granule per workgroup and iteration).
template <int P2, int NW, bool COOP>
__global__ __launch_bounds__(NW * 64) void fps_pruned_kernel(int n_cloud, int m, const float *__restrict__ dataset,
                                                          int *__restrict__ idxs, int G, int cloud0,
                                                          unsigned long long *__restrict__ gran,
                                                          int *__restrict__ err, int nc_xcd, int spin_limit)
{
    constexpr int NT = NW * 64;
    constexpr int CAP = NT * 2 * P2;
    __shared__ FpsSortLds<NW> sort_lds;
    __shared__ unsigned short order[CAP];
    __shared__ unsigned long long slot[3];
    __shared__ unsigned long long bcast[2];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // nc_xcd > 0 (COOP): workgroups are dealt to the 8 XCDs round-robin by the dispatcher, so
    // workgroup w = 8*slot + x is given slice slot%G of cloud x + 8*(slot/G): the G workgroups of
    // a cloud then share one XCD (one L2) for their per-iteration exchange.  Only a placement
    // hint: any other dispatch order is still correct.
    int cl = COOP ? (int)blockIdx.x / G : (int)blockIdx.x;
    int g = COOP ? (int)blockIdx.x % G : 0;
    if (COOP && nc_xcd > 0) {
        const int x = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
        cl = x + 8 * (slot / G);
        g = slot % G;
        if (cl >= nc_xcd) return;
    }
    const int cloud = COOP ? cloud0 + cl : cl;
    const int base = g * CAP;                                  // first point of this slice
    const int n = min(CAP, n_cloud - base);                    // points of this slice
    const float *cdata = dataset + (size_t)cloud * n_cloud * 3;   // the cloud (winner coordinates)
    const float *data = cdata + (size_t)base * 3;              // the slice (local indices)
    int *out = idxs + (size_t)cloud * m;
    unsigned long long *cgran = COOP ? gran + (size_t)cloud * 2 * G : nullptr;

    // -- grid sort of the cloud (order inside a cell is arbitrary; nothing depends on it) --
    fps_grid_sort<NW>(data, n, sort_lds, order);
    if (tid < 3) slot[tid] = 0ull;
    __syncthreads();

    // -- this lane's points: sorted rank r = ((p*16 + wave) << 7) + u*64 + lane --
    f32x2 px[P2], py[P2], pz[P2];
    int td[2 * P2], key[2 * P2];
    float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY, bz0 = INFINITY, bz1 = -INFINITY;
    // lane p < P2 holds bucket p of this wave: Tv = max running distance (bits; -1.0f = empty),
    // Nk = ~key of its holder; other lanes hold the identity of the (Tv, Nk) maximum
    int Tv = (int)0x80000000, Nk = 0;
#pragma unroll
    for (int p = 0; p < P2; ++p) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        int kmin = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = ((p * NW + wave) << 7) + u * 64 + lane;
            const bool ok = r < n;
            const int kl = ok ? (int)order[r] : 0;
            const float x = ok ? data[kl * 3 + 0] : 0.f, y = ok ? data[kl * 3 + 1] : 0.f, z = ok ? data[kl * 3 + 2] : 0.f;
            px[p][u] = x; py[p][u] = y; pz[p][u] = z;
            td[2 * p + u] = ok ? f2i(1e38f) : f2i(-1.0f);
            const int k = base + kl;                           // index in the cloud
            key[2 * p + u] = ((k & 511) << 8) | (k >> 9);
            if (ok) {
                lo[0] = fminf(lo[0], x); hi[0] = fmaxf(hi[0], x);
                lo[1] = fminf(lo[1], y); hi[1] = fmaxf(hi[1], y);
                lo[2] = fminf(lo[2], z); hi[2] = fmaxf(hi[2], z);
                kmin = min(kmin, key[2 * p + u]);
            }
        }
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
                hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
            }
        const int k0 = __builtin_amdgcn_readfirstlane(wave_min_i32_(kmin));
        if (lane == p) {
            bx0 = lo[0]; bx1 = hi[0]; by0 = lo[1]; by1 = hi[1]; bz0 = lo[2]; bz1 = hi[2];
            Tv = k0 != 0x7fffffff ? f2i(1e38f) : f2i(-1.0f);
            Nk = ~k0;
        }
    }
    // maximum of (Tv signed : Nk unsigned) over lanes 0..7 (or 0..15) -> packed slot value of this wave
    auto wave_best = [&](unsigned &hiw, unsigned &low) {
        long long v = ((long long)Tv << 32) | (unsigned)Nk;
#pragma unroll
        for (int st = 0; st < (P2 > 8 ? 4 : 3); ++st) {
            int ol, oh;
            if (st == 0) { ol = dpp_i32<0xB1>((int)v, (int)v); oh = dpp_i32<0xB1>((int)(v >> 32), (int)(v >> 32)); }
            else if (st == 1) { ol = dpp_i32<0x4E>((int)v, (int)v); oh = dpp_i32<0x4E>((int)(v >> 32), (int)(v >> 32)); }
            else if (st == 2) { ol = dpp_i32<0x141>((int)v, (int)v); oh = dpp_i32<0x141>((int)(v >> 32), (int)(v >> 32)); }
            else { ol = dpp_i32<0x140>((int)v, (int)v); oh = dpp_i32<0x140>((int)(v >> 32), (int)(v >> 32)); }
            const long long o = ((long long)oh << 32) | (unsigned)ol;
            v = o > v ? o : v;
        }
        const int bt = __builtin_amdgcn_readfirstlane((int)(v >> 32));
        low = (unsigned)__builtin_amdgcn_readfirstlane((int)v);
        hiw = bt < 0 ? 0u : (unsigned)bt + 1u;                 // fps_pack
    };
    static_assert(P2 <= 16, "wave_best reduces lanes 0..15");
    unsigned wb_hi, wb_lo;
    wave_best(wb_hi, wb_lo);
    if (tid == 0 && g == 0) out[0] = 0;
    int old = 0;                                               // wave-uniform

    for (int j = 1; j < m; ++j) {
        const float x1 = cdata[old * 3 + 0], y1 = cdata[old * 3 + 1], z1 = cdata[old * 3 + 2];
        // lower bound of the distance from the sample to the box of bucket `lane`
        const float gx = fmaxf(fmaxf(bx0 - x1, x1 - bx1), 0.f);
        const float gy = fmaxf(fmaxf(by0 - y1, y1 - by1), 0.f);
        const float gz = fmaxf(fmaxf(bz0 - z1, z1 - bz1), 0.f);
        const float lb = gx * gx + gy * gy + gz * gz;
        const unsigned dm = (unsigned)__ballot(f2i(lb) < Tv);
        if (dm) {
            bool changed = false;
#pragma unroll
            for (int p = 0; p < P2; ++p) {
                if (dm & (1u << p)) {
                    const f32x2 dx = px[p] - x1, dy = py[p] - y1, dz = pz[p] - z1;
                    const f32x2 d = dx * dx + dy * dy + dz * dz;
                    const int t0 = td[2 * p], t1 = td[2 * p + 1];
                    const int n0 = min(f2i(d[0]), t0), n1 = min(f2i(d[1]), t1);
                    td[2 * p] = n0; td[2 * p + 1] = n1;
                    const int Tp = __builtin_amdgcn_readlane(Tv, p);
                    if (__ballot((t0 == Tp && n0 != t0) || (t1 == Tp && n1 != t1))) {
                        // a holder of the bucket maximum was lowered: recompute (T, key)
                        const int v = max(n0, n1);
                        const int k0 = key[2 * p], k1 = key[2 * p + 1];
                        const int kk = n0 > n1 ? k0 : (n1 > n0 ? k1 : min(k0, k1));
                        const int wm = wave_max_i32(v);
                        const unsigned long long holders = __ballot(v == wm);
                        int wk;
                        if ((holders & (holders - 1)) == 0)
                            wk = __builtin_amdgcn_readlane(kk, __builtin_ctzll(holders));
                        else
                            wk = wave_min_i32_(v == wm ? kk : 0x7fffffff);
                        Tv = lane == p ? wm : Tv;
                        Nk = lane == p ? ~wk : Nk;
                        changed = true;
                    }
                }
            }
            if (changed) wave_best(wb_hi, wb_lo);
        }
        const int cur = j % 3;
        if (lane == 0) atomicMax(&slot[cur], ((unsigned long long)wb_hi << 32) | wb_lo);
        __syncthreads();
        unsigned long long win;
        if constexpr (!COOP) {
            win = slot[cur];
        } else {
            if (wave == 0) {
                const unsigned long long mine = slot[cur];                  // (value+1 | 0, ~key32)
                const unsigned tag = (unsigned)j & 0x7fffu;
                if (lane == 0) {
                    const unsigned long long v = (mine & 0xffffffff00000000ull) | ((mine & 0x1ffffull) << 15) | tag;
                    __hip_atomic_store(&cgran[(j & 1) * G + g], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                unsigned long long v = 0ull;
                int spins = 0;
                bool timeout = false;
                for (;;) {
                    if (lane < G)
                        v = __hip_atomic_load(&cgran[(j & 1) * G + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool ready = lane >= G || ((unsigned)v & 0x7fffu) == tag;
                    if (__all(ready)) break;
                    if (++spins > spin_limit) { timeout = true; break; }
                }
                if (timeout) {
                    if (lane == 0) { atomicExch(err, 1); bcast[j & 1] = ~0ull; }
                } else {
                    unsigned long long w = lane < G ? (v >> 15) : 0ull;
#pragma unroll
                    for (int off = 1; off < 16; off <<= 1) {
                        const unsigned long long o = __shfl_xor(w, off);
                        w = o > w ? o : w;
                    }
                    if (lane == 0) bcast[j & 1] = w;
                }
            }
            __syncthreads();
            win = bcast[j & 1];
            if (win == ~0ull) {                                          // time-out: leave (uniform)
                if (g == 0)                                              // no sample from here on: -1, never garbage
                    for (int q = j + tid; q < m; q += NT) out[q] = -1;
                return;
            }
        }
        if (tid == 0) slot[(j + 2) % 3] = 0ull;
        const unsigned k2 = ~(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)win) & 0x1ffffu;
        old = (int)((k2 >> 8) + ((k2 & 255u) << 9));
        if (tid == 0 && g == 0) out[j] = old;
    }
}

// PRCNN_FPS_PRUNE: 0 = never, 2 = whenever the kernel applies (tests), default = by size
static int fps_prune_mode()
{
    static int mode = -1;
    if (mode < 0) {
        const char *e = getenv("PRCNN_FPS_PRUNE");
        mode = e ? atoi(e) : 1;
        if (mode < 0 || mode > 2) mode = 1;
    }
    return mode;
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

// One event per device that chains the cooperative launches of this process (created on first
// use, never destroyed; hipEventDisableTiming: ordering only).
static hipEvent_t coop_chain_event()
{
    static std::mutex mu;
    static hipEvent_t ev[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!ev[dev] && hipEventCreateWithFlags(&ev[dev], hipEventDisableTiming) != hipSuccess) ev[dev] = nullptr;
    return ev[dev];
}

}  // namespace prcnn

extern "C" size_t prcnn_fps_temp_bytes(int b, int n)
{
    if (n <= 16384 || b <= 0) return 0;
    const size_t G = ((size_t)n + prcnn::COOP_SLICE - 1) / prcnn::COOP_SLICE;
    return 256 + (((size_t)b * 2 * G * sizeof(unsigned long long) + 255) & ~(size_t)255);
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
    {
        const int pm = fps_prune_mode();
        if (n <= 16384 && m >= 2 && (pm == 2 || (pm == 1 && n > 4096 && m >= 256))) {
            // 16 waves x 8 buckets; 8 waves x 16 buckets has less fixed cost per iteration but longer
            // update chains per wave: 0.82 vs 0.75 us/iteration measured
            hipLaunchKernelGGL((fps_pruned_kernel<8, 16, false>), dim3(b), dim3(1024), 0, st, n, m, inp, out, 1, 0,
                               nullptr, nullptr, 0, 0);
            PRCNN_LAUNCH_CHECK();
            return PRCNN_OK;
        }
    }
    // Up to 4096 points one 512-thread workgroup (8 waves) with up to 8 points per lane: against 1024
    // threads it halves the waves that meet at the barrier and in the LDS atomic (measured, b16:
    // 4096 -> 1024 663 -> 589 us, 1024 -> 256 140 -> 107 us, identical indices) and the registers the
    // sampler holds beside the feature kernels.
    if (n <= 512) return launch_fps<1, 1, false>(b, n, m, inp, out, st);
    if (n <= 1024) return launch_fps<2, 1, false>(b, n, m, inp, out, st);
    if (n <= 2048) return launch_fps<4, 1, false>(b, n, m, inp, out, st);
    if (n <= 4096) return launch_fps<8, 1, false>(b, n, m, inp, out, st);
    if (n <= 8192) return launch_fps<8, 2, false>(b, n, m, inp, out, st);
    if (n <= 16384) return launch_fps<16, 2, false>(b, n, m, inp, out, st);
    // without a workspace a cloud of up to 32768 points still fits one CU (min distances in LDS)
    if (!temp) return n <= 32768 ? launch_fps<32, 2, true>(b, n, m, inp, out, st) : PRCNN_WORKSPACE;
    if (m > 32767 * 2) return PRCNN_UNSUPPORTED;      // 15-bit iteration tag must not alias within reach
    // cooperative path: temp = [error word (256 B)] [b * 2 * G granules]
    const int G = (n + COOP_SLICE - 1) / COOP_SLICE;
    const size_t bytes = prcnn_fps_temp_bytes(b, n);
    if (hipMemsetAsync(temp, 0, bytes, st) != hipSuccess) return PRCNN_HIP_ERROR;
    int *err = reinterpret_cast<int *>(temp);
    unsigned long long *gran = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(temp) + 256);
    const int per_launch = 256 / G;                   // clouds whose workgroups are co-resident
    // Bound of the per-iteration wait for the other workgroups' granules (polls; ~1 us each).  The
    // default rides out seconds of other streams' kernels holding CUs; PRCNN_FPS_SPIN_LIMIT (tests)
    // forces the time-out path: 0 = give up on the first poll that finds a granule missing.
    int spin_limit = 1 << 22;
    if (const char *e = getenv("PRCNN_FPS_SPIN_LIMIT")) spin_limit = atoi(e);
    // Two cooperative launches running side by side could each hold CUs the other's unplaced
    // workgroups need (every resident workgroup spins on partners that never arrive) until both
    // time out: launches of this process are therefore chained through one event per device, so a
    // cooperative launch on one stream starts after the previous one on any stream has finished.
    // Not done under graph capture (the graph's own edges order its nodes; whoever builds a graph
    // with two cooperative samplers on parallel branches must join them).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    hipEvent_t chain = cap == hipStreamCaptureStatusNone ? coop_chain_event() : nullptr;
    if (chain && hipStreamWaitEvent(st, chain, 0) != hipSuccess) return PRCNN_HIP_ERROR;
    for (int c0 = 0; c0 < b; c0 += per_launch) {
        const int nc = b - c0 < per_launch ? b - c0 : per_launch;
        if (fps_prune_mode() != 0 && m >= 256) {    // long runs: pruned slices (in-workgroup sort pays off)
            static const int xcd = getenv("PRCNN_FPS_XCD") ? atoi(getenv("PRCNN_FPS_XCD")) : 1;   // 0: tuning aid
            if (xcd)
                hipLaunchKernelGGL((fps_pruned_kernel<8, 16, true>), dim3(8 * G * ((nc + 7) / 8)), dim3(1024), 0, st,
                                   n, m, inp, out, G, c0, gran, err, nc, spin_limit);
            else
                hipLaunchKernelGGL((fps_pruned_kernel<8, 16, true>), dim3(nc * G), dim3(1024), 0, st, n, m, inp, out,
                                   G, c0, gran, err, 0, spin_limit);
        }
        else
            hipLaunchKernelGGL(fps_coop_kernel, dim3(nc * G), dim3(1024), 0, st, n, m, G, c0, inp, gran, err, out,
                               spin_limit);
        PRCNN_LAUNCH_CHECK();
    }
    if (chain && hipEventRecord(chain, st) != hipSuccess) return PRCNN_HIP_ERROR;
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
