// three_nn (+ fused interpolation) with spatial culling, for large n x m.
//
// Same contract and bit-identical results as prcnn_three_nn / prcnn_three_nn_interpolate
// (reference: threenn_cpu, models/tf_ops/3d_interpolation/tf_interpolate.cpp:60-103), but
// the b*n*m brute-force tests are cut by an order of magnitude:
//   1. per cloud, queries and known points are bucketed into a uniform Morton-ordered grid
//      over their joint bounding box (one workgroup per set: counting sort in LDS);
//   2. the sorted known points are cut into tiles of 64 with a bounding box each;
//   3. a wave owns 64 consecutive sorted queries (a compact box); it visits known tiles in
//      order of increasing box-to-box distance and stops when the nearest unvisited tile is
//      farther than the worst third-best distance of its 64 queries.
// Exactness: the box-to-box bound is evaluated with the same unfused fp32 operations as
// the point distance ((dx*dx+dy*dy)+dz*dz); every operation is monotone under rounding,
// so the bound never exceeds the computed distance of any pair it covers, and a tile is
// skipped only when bound > worst (strictly).  The reference's strict-'<' insertion in
// index order equals "three smallest by (distance, index)"; since tiles are not visited
// in index order, insertion here compares (distance, index) pairs lexicographically.
#include "common.h"
#include "ball.h"

namespace prcnn {

#ifndef NS_NW
#define NS_NW 4
#endif
constexpr int NS_THREADS = 64 * NS_NW;      // search kernel: NS_NW waves share one block of 64 queries
constexpr int NS_KEY_BITS = 15;           // grid cells per cloud = 32768 (128 KB LDS histogram)
constexpr int NS_CELLS = 1 << NS_KEY_BITS;

// ---- 1+2. per-cloud counting sort of one point set by grid cell, entirely in LDS -----------
// grid (2, b): blockIdx.x = 0 sorts the queries, 1 the known points.  Both workgroups derive
// the same grid from the joint bounding box: a uniform cell size, doubled until the cell
// coordinates of the three axes fit NS_KEY_BITS bits together (a flat KITTI scene gets
// e.g. 7+2+6 bits, a cubic cloud 5+5+5), interleaved low bits first (Morton-like order).
// The order inside a cell is whatever the LDS atomics produce: results do not depend on it.
// REG: the workgroup's own point set (<= 16384 points) is read from memory ONCE, 16 points per
// thread with all loads in flight together, and stays in registers for the box, the histogram
// and the scatter pass (the looping form reads every point three times, one dependent load
// latency per iteration: 56 us -> see DESIGN.md 4.4).
template <bool REG>
__global__ __launch_bounds__(1024) void ns_sort_kernel(int n, int m, const float *__restrict__ xyz1,
                                                       const float *__restrict__ xyz2,
                                                       int *__restrict__ list_q, int *__restrict__ list_k)
{
    constexpr int PPT = 16;
    __shared__ __attribute__((aligned(16))) int hist[NS_CELLS];
    __shared__ float red[16][6];
    __shared__ float sbox[4];
    __shared__ int sbits[3];
    __shared__ int wsum[16];
    const int bi = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *a = xyz1 + (size_t)bi * n * 3;
    const float *k = xyz2 + (size_t)bi * m * 3;
    const bool known = blockIdx.x != 0;
    const float *pts = known ? k : a;                  // the set this workgroup sorts
    const float *oth = known ? a : k;                  // the other set: joint bounding box only
    const int np = known ? m : n, no = known ? n : m;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float px[REG ? PPT : 1], py[REG ? PPT : 1], pz[REG ? PPT : 1];
    if (REG) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            const int i = tid + q * 1024;
            const bool in = i < np;
            px[q] = in ? pts[(size_t)i * 3 + 0] : __builtin_nanf("");      // NaN: ignored by fmin / fmax
            py[q] = in ? pts[(size_t)i * 3 + 1] : __builtin_nanf("");
            pz[q] = in ? pts[(size_t)i * 3 + 2] : __builtin_nanf("");
        }
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            lo[0] = fminf(lo[0], px[q]); hi[0] = fmaxf(hi[0], px[q]);
            lo[1] = fminf(lo[1], py[q]); hi[1] = fmaxf(hi[1], py[q]);
            lo[2] = fminf(lo[2], pz[q]); hi[2] = fmaxf(hi[2], pz[q]);
        }
    }
    {
        // box of the other set (REG) or of both sets (looping form): four points per trip, so four
        // loads per coordinate are in flight
        const int tot = REG ? no : n + m;
        for (int i0 = tid; i0 < tot; i0 += 4 * 1024) {
            float v[4][3];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 1024;
                const float *p = REG ? oth + (size_t)i * 3 : (i < n ? a + (size_t)i * 3 : k + (size_t)(i - n) * 3);
#pragma unroll
                for (int d = 0; d < 3; ++d) v[u][d] = i < tot ? p[d] : __builtin_nanf("");
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int d = 0; d < 3; ++d) {
                    lo[d] = fminf(lo[d], v[u][d]);          // NaN coordinates are ignored by fmin/fmax
                    hi[d] = fmaxf(hi[d], v[u][d]);
                }
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
        }
    if (lane == 0)
        for (int d = 0; d < 3; ++d) { red[wave][d] = lo[d]; red[wave][3 + d] = hi[d]; }
    for (int c = tid; c < NS_CELLS; c += 1024) hist[c] = 0;
    __syncthreads();
    if (tid == 0) {
        float blo[3], ext[3], mx = 0.f;
        for (int d = 0; d < 3; ++d) {
            float l = red[0][d], h = red[0][3 + d];
            for (int w = 1; w < 16; ++w) { l = fminf(l, red[w][d]); h = fmaxf(h, red[w][3 + d]); }
            blo[d] = l;
            ext[d] = h - l;
            if (!(ext[d] >= 0.f) || !(ext[d] < INFINITY)) ext[d] = 0.f;      // empty / non-finite axis
            mx = fmaxf(mx, ext[d]);
            sbox[d] = l;
        }
        int bits[3] = {0, 0, 0};
        float inv = 0.f;
        // (an extent so small that mx / 1024 underflows to zero keeps every point in cell 0: the loop
        // below would never end on cell == 0)
        if (mx > 0.f && mx / 1024.f > 0.f) {
            float cell = mx / 1024.f;                 // at most 10 bits on the longest axis
            for (;;) {
                int tot = 0;
                for (int d = 0; d < 3; ++d) {
                    const float cells = ext[d] / cell;
                    int bt = 0;
                    while (bt < 10 && (float)(1 << bt) < cells) ++bt;
                    bits[d] = bt;
                    tot += bt;
                }
                if (tot <= NS_KEY_BITS) break;
                cell *= 2.f;
            }
            inv = 1.f / cell;
        }
        sbox[3] = inv;
        for (int d = 0; d < 3; ++d) sbits[d] = bits[d];
    }
    __syncthreads();
    const float b0 = sbox[0], b1 = sbox[1], b2 = sbox[2], inv = sbox[3];
    const int t0 = sbits[0], t1 = sbits[1], t2 = sbits[2];
    int *list = known ? list_k + (size_t)bi * m : list_q + (size_t)bi * n;
    // bit-interleaved key through three tables (one per axis: the cell coordinate's bits spread to
    // their key positions; built once by the workgroup, sharing the histogram's barrier): the
    // 30-step interleave per point was the kernel's largest cost
    __shared__ int spread[3][1024];
    {
        const int q = tid;                                   // 1024 threads: one table entry per axis each
        int k0 = 0, k1 = 0, k2 = 0, pos = 0;
        for (int lv = 0; lv < 10; ++lv) {
            if (lv < t0) { k0 |= ((q >> lv) & 1) << pos; ++pos; }
            if (lv < t1) { k1 |= ((q >> lv) & 1) << pos; ++pos; }
            if (lv < t2) { k2 |= ((q >> lv) & 1) << pos; ++pos; }
        }
        spread[0][q] = k0; spread[1][q] = k1; spread[2][q] = k2;
    }
    __syncthreads();
    auto key_xyz = [&](float x, float y, float z) {
        const float f0 = (x - b0) * inv, f1 = (y - b1) * inv, f2 = (z - b2) * inv;
        const int q0 = f0 > 0.f ? (int)fminf(f0, (float)((1 << t0) - 1)) : 0;      // NaN -> 0
        const int q1 = f1 > 0.f ? (int)fminf(f1, (float)((1 << t1) - 1)) : 0;
        const int q2 = f2 > 0.f ? (int)fminf(f2, (float)((1 << t2) - 1)) : 0;
        return spread[0][q0] | spread[1][q1] | spread[2][q2];
    };
    auto key_of = [&](int i) {
        const float *p = pts + (size_t)i * 3;
        return key_xyz(p[0], p[1], p[2]);
    };
    int keys[REG ? PPT : 1];
    if (REG) {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            keys[q] = key_xyz(px[q], py[q], pz[q]);
            if (tid + q * 1024 < np) atomicAdd(&hist[keys[q]], 1);
        }
    } else {
        for (int i = tid; i < np; i += 1024) atomicAdd(&hist[key_of(i)], 1);
    }
    __syncthreads();
    // exclusive scan of the histogram in place: 32 consecutive bins per thread, moved as 8 x 16 bytes
    constexpr int PER = NS_CELLS / 1024;
    static_assert(PER == 32, "scan layout");
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    i32x4 hv[PER / 4];
    int sum = 0;
#pragma unroll
    for (int c = 0; c < PER / 4; ++c) {
        hv[c] = *reinterpret_cast<const i32x4 *>(&hist[tid * PER + c * 4]);
        sum += hv[c][0] + hv[c][1] + hv[c][2] + hv[c][3];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int c = 0; c < PER / 4; ++c) {
        i32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = run; run += hv[c][e]; }
        *reinterpret_cast<i32x4 *>(&hist[tid * PER + c * 4]) = o;
    }
    __syncthreads();
    if (REG) {
#pragma unroll
        for (int q = 0; q < PPT; ++q)
            if (tid + q * 1024 < np) list[atomicAdd(&hist[keys[q]], 1)] = tid + q * 1024;
    } else {
        for (int i = tid; i < np; i += 1024) list[atomicAdd(&hist[key_of(i)], 1)] = i;
    }
}

static void launch_ns_sort(int b, int n, int m, const float *xyz1, const float *xyz2, int *list_q, int *list_k,
                           hipStream_t st)
{
    if (n <= 16384 && m <= 16384)
        hipLaunchKernelGGL(ns_sort_kernel<true>, dim3(2, b), dim3(1024), 0, st, n, m, xyz1, xyz2, list_q, list_k);
    else
        hipLaunchKernelGGL(ns_sort_kernel<false>, dim3(2, b), dim3(1024), 0, st, n, m, xyz1, xyz2, list_q, list_k);
}

// ---- 3. sorted known points (x, y, z, original index) and their tile boxes -----------------
// one wave per tile of 64 sorted known points
template <bool PAIR>
__global__ __launch_bounds__(256) void ns_tile_kernel(int m, int tiles_per_cloud, long long total_tiles,
                                                      const float *__restrict__ xyz2,
                                                      const int *__restrict__ sorted,     // (b, m)
                                                      float4 *__restrict__ kp,            // (b, tiles*64)
                                                      float *__restrict__ tb)             // (b, tiles, 8)
{
    const int lane = threadIdx.x & 63;
    const long long t = blockIdx.x * 4ll + (threadIdx.x >> 6);
    if (t >= total_tiles) return;
    const long long bi = t / tiles_per_cloud;
    const int tl = (int)(t - bi * tiles_per_cloud);
    const int s = tl * 64 + lane;
    const float nan = __builtin_nanf("");
    float4 p = make_float4(nan, nan, nan, 0.f);
    if (s < m) {
        const int k = sorted[bi * m + s];
        const float *q = xyz2 + ((size_t)bi * m + k) * 3;
        p = make_float4(q[0], q[1], q[2], __builtin_bit_cast(float, k));
    }
    // pair layout: entry 2j = (x_2j, x_2j+1, y_2j, y_2j+1), entry 2j+1 = (z_2j, z_2j+1, idx_2j, idx_2j+1),
    // so that the search runs two candidates through every packed-fp32 operation
    {
        const float ox = __shfl_xor(p.x, 1), oy = __shfl_xor(p.y, 1), oz = __shfl_xor(p.z, 1), ow = __shfl_xor(p.w, 1);
        const float4 e = (lane & 1) ? make_float4(oz, p.z, ow, p.w) : make_float4(p.x, ox, p.y, oy);
        kp[bi * tiles_per_cloud * 64 + s] = PAIR ? e : p;        // !PAIR: (x, y, z, idx) per point
    }
    float lo[3] = {p.x, p.y, p.z}, hi[3] = {p.x, p.y, p.z};
    if (!(s < m)) {
        for (int d = 0; d < 3; ++d) { lo[d] = INFINITY; hi[d] = -INFINITY; }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
        }
    if (lane < 6) {
        float v = lane == 0 ? lo[0] : lane == 1 ? lo[1] : lane == 2 ? lo[2] : lane == 3 ? hi[0] : lane == 4 ? hi[1] : hi[2];
        tb[t * 8 + lane] = v;
    }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// two candidates (pair layout of ns_tile_kernel) against one query; each component is the
// scalar computation of tf_interpolate.cpp:73, unfused
__device__ __forceinline__ f32x2 sqd2(const float4 a, const float4 b, f32x2 qx, f32x2 qy, f32x2 qz)
{
    const f32x2 px = {a.x, a.y}, py = {a.z, a.w}, pz = {b.x, b.y};
    const f32x2 dx = px - qx, dy = py - qy, dz = pz - qz;
    return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ float wave_minf(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_maxf(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// pointnet_util.py:211-214
__device__ __forceinline__ void fp_weights3s(float d0, float d1, float d2, float &w0, float &w1, float &w2)
{
    d0 = fmaxf(d0, 1e-10f); d1 = fmaxf(d1, 1e-10f); d2 = fmaxf(d2, 1e-10f);
    const float r0 = 1.0f / d0, r1 = 1.0f / d1, r2 = 1.0f / d2;
    const float norm = (r0 + r1) + r2;
    w0 = r0 / norm; w1 = r1 / norm; w2 = r2 / norm;
}

// ---- 4. search (+ optional fused interpolation) --------------------------------------------
// A workgroup owns NS_QPB = 64 consecutive sorted queries; its FOUR waves all hold the same 64 queries
// (one per lane) and share the tiles among them: wave w visits tiles t with t % 4 == w, nearest
// first, keeping its own top three.  The waves publish their current third-best distance per query
// in LDS; the minimum over the four is an upper bound of the final third-best (each wave's list is a
// subset of all candidates), so every wave prunes tiles and screens candidates with it.  The final
// three are the three smallest (distance, index) keys of the union of the four lists (merge below).
// Why: with one wave per 64 queries, the cost of a wave spreads 4-5x around its mean (6.7 tiles and 88
// insertions on average, 25 and 430 at worst, RPN level 1) and all waves are resident at once, so the
// kernel lasted as long as its slowest wave: 98 us at an average wave residency of 28 us.  Four
// waves per block cut the longest dependent chain by four and give the dispatcher four rounds of
// workgroups to balance.
constexpr int NS_QPB = 64;

template <int VEC, bool INTERP>
__global__ __launch_bounds__(NS_THREADS) void ns_search_kernel(
    int n, int m, int c, int tiles_per_cloud, const float *__restrict__ xyz1,
    const int *__restrict__ qsorted,            // (b, n) query indices in Morton order
    const float4 *__restrict__ kp, const float *__restrict__ tb, const float *__restrict__ points2,
    float *__restrict__ out, long long ldo, float *__restrict__ dist, int *__restrict__ idx, int epi,
    EpiAffine aff)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ float s_d3[NS_THREADS / 64][NS_QPB];
    __shared__ unsigned long long s_key[NS_THREADS / 64][3][NS_QPB];
    __shared__ int s_q[NS_QPB];
    __shared__ int s_i[NS_QPB * 3];
    __shared__ float s_w[NS_QPB * 3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bi = blockIdx.y;
    const int j0 = blockIdx.x * NS_QPB;
    const int js = j0 + lane;                      // position in sorted order
    const bool ok = js < n;
    const int qi = qsorted[(size_t)bi * n + (ok ? js : n - 1)];
    const float *q = xyz1 + ((size_t)bi * n + qi) * 3;
    const float x1 = q[0], y1 = q[1], z1 = q[2];
    // the block's query box (lanes past n repeat the last query: no effect on the box)
    const float qlo0 = wave_minf(x1), qlo1 = wave_minf(y1), qlo2 = wave_minf(z1);
    const float qhi0 = wave_maxf(x1), qhi1 = wave_maxf(y1), qhi2 = wave_maxf(z1);
    const float4 *kpc = kp + (size_t)bi * tiles_per_cloud * 64;
    const float *tbc = tb + (size_t)bi * tiles_per_cloud * 8;
    const f32x2 qx = {x1, x1}, qy = {y1, y1}, qz = {z1, z1};
    Top3K t = top3k_init();
    // published third-best distances: read without ordering (any value ever published is a valid
    // bound; they only shrink)
    volatile float *vd3 = &s_d3[0][0];
    vd3[wave * NS_QPB + lane] = INFINITY;
    __syncthreads();
    float sd = INFINITY;                           // screening bound: min over the four waves
    for (int tbase = 0; tbase < tiles_per_cloud; tbase += 64) {
        // lane l: box of tile tbase + l and the lower bound between it and the block's query box
        const int nt = min(64, tiles_per_cloud - tbase);
        const bool mine = lane < nt && (lane % NS_NW) == wave;    // tbase % NS_NW == 0
        float lb = INFINITY;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, b4 = 0.f, b5 = 0.f;
        if (mine) {
            const float *bx = tbc + (size_t)(tbase + lane) * 8;
            b0 = bx[0]; b1 = bx[1]; b2 = bx[2]; b3 = bx[3]; b4 = bx[4]; b5 = bx[5];
            const float g0 = fmaxf(fmaxf(b0 - qhi0, qlo0 - b3), 0.f);
            const float g1 = fmaxf(fmaxf(b1 - qhi1, qlo1 - b4), 0.f);
            const float g2 = fmaxf(fmaxf(b2 - qhi2, qlo2 - b5), 0.f);
            lb = g0 * g0 + g1 * g1 + g2 * g2;                // same op order as sqd2(): unfused
            if (!(lb == lb)) lb = 0.f;                        // NaN box: never skip
        }
        int lbi = mine ? f2i(lb) : 0x7fffffff;                // lb >= 0: int order = float order
        // Next tile worth a visit, nearest first, or -1.  The walk over this group of tiles ends
        // when the nearest unvisited tile is farther (box to box) than the worst bound of the 64
        // queries.  A tile inside that bound is still skipped when NO lane can use it: its own
        // point-to-box bound (same unfused operations, monotone under rounding, so never above the
        // computed distance of a pair it covers) exceeds its own bound.
        auto next_tile = [&]() -> int {
#pragma unroll
            for (int w = 0; w < NS_NW; ++w) sd = fminf(sd, vd3[w * NS_QPB + lane]);
            const int worst = wave_max_i32(ok ? f2i(sd) : 0);         // distances >= +0: int order
            for (;;) {
                const int mn = -wave_max_i32(-lbi);
                if (mn == 0x7fffffff) return -1;
                if (mn > worst) return -1;                            // every remaining tile is farther
                const unsigned long long who = __ballot(lbi == mn);
                const int tl = __builtin_ctzll(who);
                if (lane == tl) lbi = 0x7fffffff;
                auto rl = [&](float v) { return i2f(__builtin_amdgcn_readlane(f2i(v), tl)); };   // bits, not value
                const float t0 = rl(b0), t1 = rl(b1), t2 = rl(b2), t3 = rl(b3), t4 = rl(b4), t5 = rl(b5);
                const float g0 = fmaxf(fmaxf(t0 - x1, x1 - t3), 0.f);
                const float g1 = fmaxf(fmaxf(t1 - y1, y1 - t4), 0.f);
                const float g2 = fmaxf(fmaxf(t2 - z1, z1 - t5), 0.f);
                const float ll = g0 * g0 + g1 * g1 + g2 * g2;
                // '<=': an equal distance with a smaller index must still get in; NaN: visit
                if (__ballot(ok && !(ll > sd))) return tl;
            }
        };
        // The tile's 64 points are the same for every lane: they are read through the scalar cache
        // (wave-uniform address into memory this kernel never writes) and enter the packed operations
        // as scalar operands -- no LDS staging, no LDS bandwidth (four waves per SIMD reading 64 B per
        // lane per step through the CU's one LDS port was the kernel's bound).
        for (int tl = next_tile(); tl >= 0; tl = next_tile()) {
            const float4 *kt = kpc + (size_t)(tbase + tl) * 64;
#pragma unroll 4
            for (int k = 0; k < 64; k += 4) {
                const float4 a0 = kt[k], b0_ = kt[k + 1], a1 = kt[k + 2], b1_ = kt[k + 3];
                const f32x2 da = sqd2(a0, b0_, qx, qy, qz), db = sqd2(a1, b1_, qx, qy, qz);
                const float mn4 = fminf(fminf(da[0], da[1]), fminf(db[0], db[1]));
                // '<=': tiles are not visited in index order, an equal distance with a smaller
                // index must still get in (the 64-bit key decides)
                if (__builtin_amdgcn_ballot_w64(mn4 <= sd)) {
                    if (__builtin_amdgcn_ballot_w64(da[0] <= sd)) { top3k_insert(t, da[0], f2i(b0_.z)); sd = fminf(sd, t.d3); }
                    if (__builtin_amdgcn_ballot_w64(da[1] <= sd)) { top3k_insert(t, da[1], f2i(b0_.w)); sd = fminf(sd, t.d3); }
                    if (__builtin_amdgcn_ballot_w64(db[0] <= sd)) { top3k_insert(t, db[0], f2i(b1_.z)); sd = fminf(sd, t.d3); }
                    if (__builtin_amdgcn_ballot_w64(db[1] <= sd)) { top3k_insert(t, db[1], f2i(b1_.w)); sd = fminf(sd, t.d3); }
                }
            }
            vd3[wave * NS_QPB + lane] = t.d3;
        }
    }
    // merge: the three smallest keys of the four lists (a candidate some wave screened out lies above
    // the final third-best; the final three each sit in the list of the wave that met them)
    s_key[wave][0][lane] = t.k1; s_key[wave][1][lane] = t.k2; s_key[wave][2][lane] = t.k3;
    __syncthreads();
    if (wave != 0) {
        if (!INTERP) return;
    } else {
#pragma unroll
        for (int w = 1; w < NS_THREADS / 64; ++w)
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const unsigned long long k = s_key[w][r][lane];
                top3k_insert(t, top3k_dist(k), top3k_index(k));
            }
        if (ok) {
            const size_t o = ((size_t)bi * n + qi) * 3;
            if (dist) { dist[o] = top3k_dist(t.k1); dist[o + 1] = top3k_dist(t.k2); dist[o + 2] = top3k_dist(t.k3); }
            if (idx) { idx[o] = top3k_index(t.k1); idx[o + 1] = top3k_index(t.k2); idx[o + 2] = top3k_index(t.k3); }
        }
        if (!INTERP) return;
        float w0, w1, w2;
        fp_weights3s(top3k_dist(t.k1), top3k_dist(t.k2), top3k_dist(t.k3), w0, w1, w2);
        s_q[lane] = qi;
        s_i[lane * 3] = top3k_index(t.k1); s_i[lane * 3 + 1] = top3k_index(t.k2); s_i[lane * 3 + 2] = top3k_index(t.k3);
        s_w[lane * 3] = w0; s_w[lane * 3 + 1] = w1; s_w[lane * 3 + 2] = w2;
    }
    __syncthreads();
    const int rows = min(NS_QPB, n - j0);
    const int cv = c / VEC;
    const float *pb = points2 + (size_t)bi * m * c;
    if (cv <= NS_THREADS && NS_THREADS % cv == 0) {
        // The thread's column group is the same for every row it handles (NS_THREADS is a multiple of
        // c / VEC): the epilogue's per-column operands (t, the rows of u) are loaded ONCE, and the
        // loop body is three gathered loads, the weighted sum, the narrow p1 product and one store
        // -- the general form below re-reads 1 + 5 c1 values per element and made this phase the
        // longer half of the fused kernel.
        const int col = (tid % cv) * VEC;
        const int rstep = NS_THREADS / cv;
        float tv[VEC], uv[4][VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            tv[q] = epi >= 3 ? aff.t[col + q] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) uv[k][q] = (epi >= 3 && k < aff.c1) ? aff.u[(size_t)k * c + col + q] : 0.f;
        }
        for (int r = tid / cv; r < rows; r += rstep) {
            const size_t qrow = (size_t)bi * n + s_q[r];
            const vec_t a = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3] * c + col);
            const vec_t bb = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3 + 1] * c + col);
            const vec_t cc = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3 + 2] * c + col);
            vec_t v = a * s_w[r * 3] + bb * s_w[r * 3 + 1] + cc * s_w[r * 3 + 2];   // unfused
            vec_t *op = reinterpret_cast<vec_t *>(out + qrow * ldo + col);
            if (epi >= 3) {                              // 3: affine + relu;  4: affine (EpiAffine)
                // same operation order as epi_affine_add: x = t; x += p1[k] * u[k] (k ascending); v += x
                float x[VEC];
#pragma unroll
                for (int q = 0; q < VEC; ++q) x[q] = tv[q];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < aff.c1) {
                        const float pv = aff.p1[qrow * aff.c1 + k];
#pragma unroll
                        for (int q = 0; q < VEC; ++q) x[q] += pv * uv[k][q];
                    }
#pragma unroll
                for (int q = 0; q < VEC; ++q) { v[q] += x[q]; if (epi == 3) v[q] = fmaxf(v[q], 0.f); }
            } else if (epi) {                            // 1: out += v;  2: out = relu(out + v)
                v = *op + v;
                if (epi == 2) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) v[q] = fmaxf(v[q], 0.f);
                }
            }
            *op = v;
        }
        return;
    }
    for (int e = tid; e < rows * cv; e += NS_THREADS) {
        const int r = e / cv;
        const int col = (e - r * cv) * VEC;
        const vec_t a = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3] * c + col);
        const vec_t bb = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3 + 1] * c + col);
        const vec_t cc = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3 + 2] * c + col);
        vec_t v = a * s_w[r * 3] + bb * s_w[r * 3 + 1] + cc * s_w[r * 3 + 2];   // unfused
        vec_t *op = reinterpret_cast<vec_t *>(out + ((size_t)bi * n + s_q[r]) * ldo + col);
        if (epi >= 3) {                              // 3: affine + relu;  4: affine (EpiAffine)
            float vv[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) vv[q] = v[q];
            epi_affine_add<VEC>(aff, (size_t)bi * n + s_q[r], c, col, vv);
#pragma unroll
            for (int q = 0; q < VEC; ++q) v[q] = epi == 3 ? fmaxf(vv[q], 0.f) : vv[q];
        } else if (epi) {                            // 1: out += v;  2: out = relu(out + v)
            v = *op + v;
            if (epi == 2) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) v[q] = fmaxf(v[q], 0.f);
            }
        }
        *op = v;
    }
}

// ---- workspace ---------------------------------------------------------------------------
struct NsWs {
    int *list_q;      // (b, n) query indices in grid order
    int *list_k;      // (b, m)
    float4 *kp;       // (b, tiles*64)
    float *tb;        // (b, tiles, 8)
    size_t bytes;
};

static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

static NsWs ns_carve(void *ws, long long b, long long n, long long m)
{
    const long long tiles = (m + 63) / 64;
    size_t o = 0;
    const size_t o_q = o; o += up256((size_t)b * n * 4);
    const size_t o_k = o; o += up256((size_t)b * m * 4);
    const size_t o_p = o; o += up256((size_t)b * tiles * 64 * 16);
    const size_t o_t = o; o += up256((size_t)b * tiles * 8 * 4);
    NsWs w = {};
    w.bytes = o;
    if (ws) {                                   // sizes only when called without a buffer
        char *p = reinterpret_cast<char *>(ws);
        w.list_q = reinterpret_cast<int *>(p + o_q);
        w.list_k = reinterpret_cast<int *>(p + o_k);
        w.kp = reinterpret_cast<float4 *>(p + o_p);
        w.tb = reinterpret_cast<float *>(p + o_t);
    }
    return w;
}

// order_q / order_k: grid orders of xyz1 / xyz2 computed beforehand (prcnn_grid_order on the same two sets,
// in either role): the sort launch is skipped
static int ns_run(int b, int n, int m, int c, const float *xyz1, const float *xyz2, const float *points2,
                  float *out, long long ldo, float *dist, int *idx, void *ws, size_t ws_bytes, hipStream_t st,
                  int epi = 0, EpiAffine aff = EpiAffine{}, const int *order_q = nullptr, const int *order_k = nullptr)
{
    NsWs w = ns_carve(ws, b, n, m);
    if (!ws || ws_bytes < w.bytes) return PRCNN_WORKSPACE;
    if (((uintptr_t)ws & 255) != 0) return PRCNN_BAD_ARG;
    if ((order_q == nullptr) != (order_k == nullptr)) return PRCNN_BAD_ARG;
    const int tiles = (m + 63) / 64;
    if (order_q) { w.list_q = const_cast<int *>(order_q); w.list_k = const_cast<int *>(order_k); }
    else launch_ns_sort(b, n, m, xyz1, xyz2, w.list_q, w.list_k, st);
    const long long total_tiles = (long long)b * tiles;
    hipLaunchKernelGGL(ns_tile_kernel<true>, dim3((unsigned)((total_tiles + 3) / 4)), dim3(256), 0, st, m, tiles,
                       total_tiles, xyz2, w.list_k, w.kp, w.tb);
    dim3 grid((n + NS_QPB - 1) / NS_QPB, b);
    if (!points2) {
        hipLaunchKernelGGL((ns_search_kernel<1, false>), grid, dim3(NS_THREADS), 0, st, n, m, 0, tiles, xyz1,
                           w.list_q, w.kp, w.tb, nullptr, nullptr, 0ll, dist, idx, 0, aff);
    } else {
        const bool a16 = (((uintptr_t)points2 | (uintptr_t)out) & 15) == 0 && ldo % 4 == 0;
        if (c % 4 == 0 && a16)
            hipLaunchKernelGGL((ns_search_kernel<4, true>), grid, dim3(NS_THREADS), 0, st, n, m, c, tiles, xyz1,
                               w.list_q, w.kp, w.tb, points2, out, ldo, dist, idx, epi, aff);
        else
            hipLaunchKernelGGL((ns_search_kernel<1, true>), grid, dim3(NS_THREADS), 0, st, n, m, c, tiles, xyz1,
                               w.list_q, w.kp, w.tb, points2, out, ldo, dist, idx, epi, aff);
    }
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

// ---- 5. ball query over the same grid ---------------------------------------------------------
// query_ball_point (tf_grouping_g.cu:3-36): the first nsample point indices, in index order,
// with max(sqrt(d2),1e-20) < radius, i.e. d2 < thr2 (ball_threshold).  Here the POINTS are
// bucketed and tiled; a wave owns four neighbouring centroids and visits only the tiles whose
// box comes closer than the largest radius to the box of its centroids (same rounding-monotone
// bound as above: a skipped tile cannot contain a hit).  Hits arrive in arbitrary index order,
// so every (centroid, scale) keeps a small LDS buffer that is reduced to its nsample smallest
// indices, ascending (rank counting), whenever it would overflow and once at the end:
// identical rows, padding and pts_cnt as the index-order scan.
constexpr int BC_CPW = 4;
constexpr int BC_BUF = 128;

// keep the min(L, ns) smallest of the L (<= 128, distinct) entries of b, ascending
__device__ __forceinline__ int bc_compact(int *b, int L, int ns, int lane)
{
    const int e0 = lane < L ? b[lane] : 0x7fffffff;
    const int e1 = lane + 64 < L ? b[lane + 64] : 0x7fffffff;
    int r0 = 0, r1 = 0;
    for (int u = 0; u < L; ++u) {
        const int v = b[u];
        r0 += v < e0 ? 1 : 0;
        r1 += v < e1 ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane < L && r0 < ns) b[r0] = e0;
    if (lane + 64 < L && r1 < ns) b[r1] = e1;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return min(L, ns);
}

template <int NS>
__global__ __launch_bounds__(256) void bq_culled_kernel(int n, int m, int tiles, const BallScales sc,
                                                        const float *__restrict__ xyz2,
                                                        const int *__restrict__ csorted,   // (b, m)
                                                        const float4 *__restrict__ pp,     // (b, tiles*64)
                                                        const float *__restrict__ tb)      // (b, tiles, 8)
{
    __shared__ int buf[4][BC_CPW][NS][BC_BUF];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bi = blockIdx.y;
    const int j0 = (blockIdx.x * 4 + wave) * BC_CPW;
    if (j0 >= m) return;
    int cj[BC_CPW];
    float cx[BC_CPW], cy[BC_CPW], cz[BC_CPW];
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int c = 0; c < BC_CPW; ++c) {
        const int js = min(j0 + c, m - 1);                   // past the end: repeat the last one
        cj[c] = j0 + c < m ? csorted[(size_t)bi * m + js] : -1;
        const float *p = xyz2 + ((size_t)bi * m + csorted[(size_t)bi * m + js]) * 3;
        cx[c] = p[0]; cy[c] = p[1]; cz[c] = p[2];
        clo[0] = fminf(clo[0], cx[c]); chi[0] = fmaxf(chi[0], cx[c]);
        clo[1] = fminf(clo[1], cy[c]); chi[1] = fmaxf(chi[1], cy[c]);
        clo[2] = fminf(clo[2], cz[c]); chi[2] = fmaxf(chi[2], cz[c]);
    }
    float thrmax = sc.thr2[0];
#pragma unroll
    for (int s = 1; s < NS; ++s) thrmax = fmaxf(thrmax, sc.thr2[s]);
    int cnt[BC_CPW][NS], tot[BC_CPW][NS];
#pragma unroll
    for (int c = 0; c < BC_CPW; ++c)
#pragma unroll
        for (int s = 0; s < NS; ++s) { cnt[c][s] = 0; tot[c][s] = 0; }
    const unsigned long long lt = (1ull << lane) - 1ull;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float4 *ppc = pp + (size_t)bi * tiles * 64;
    const float *tbc = tb + (size_t)bi * tiles * 8;
    for (int tbase = 0; tbase < tiles; tbase += 64) {
        const int nt = min(64, tiles - tbase);
        bool cand = false;
        if (lane < nt) {
            const float *bx = tbc + (size_t)(tbase + lane) * 8;
            const float g0 = fmaxf(fmaxf(bx[0] - chi[0], clo[0] - bx[3]), 0.f);
            const float g1 = fmaxf(fmaxf(bx[1] - chi[1], clo[1] - bx[4]), 0.f);
            const float g2 = fmaxf(fmaxf(bx[2] - chi[2], clo[2] - bx[5]), 0.f);
            const float lb = g0 * g0 + g1 * g1 + g2 * g2;    // same op order as the distance: unfused
            cand = !(lb >= thrmax);                          // NaN bound: visit
        }
        unsigned long long todo = __ballot(cand);
        // one tile ahead: the next candidate tile's points are on their way while this one is tested
        // (the walk is a chain of dependent memory round trips otherwise: a wave does ~40 VALU
        // instructions per tile)
        float4 pn = make_float4(0.f, 0.f, 0.f, 0.f);
        if (todo) pn = ppc[(size_t)(tbase + __builtin_ctzll(todo)) * 64 + lane];
        while (todo) {
            todo &= todo - 1;
            const float4 p = pn;                                      // NaN padding never hits
            if (todo) pn = ppc[(size_t)(tbase + __builtin_ctzll(todo)) * 64 + lane];
            const int k = f2i(p.w);
            const f32x2 px = {p.x, p.x}, py = {p.y, p.y}, pz = {p.z, p.z};
#pragma unroll
            for (int c0 = 0; c0 < BC_CPW; c0 += 2) {
                const f32x2 ccx = {cx[c0], cx[c0 + 1]}, ccy = {cy[c0], cy[c0 + 1]}, ccz = {cz[c0], cz[c0 + 1]};
                const f32x2 dx = ccx - px, dy = ccy - py, dz = ccz - pz;
                const f32x2 d2p = dx * dx + dy * dy + dz * dz;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = c0 + h;
                    if (__builtin_amdgcn_ballot_w64(d2p[h] < thrmax) == 0) continue;
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        const bool hit = d2p[h] < sc.thr2[s];
                        const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
                        if (hm == 0) continue;
                        const int nh = __builtin_popcountll(hm);
                        int *bb = buf[wave][c][s];
                        if (cnt[c][s] + nh > BC_BUF) cnt[c][s] = bc_compact(bb, cnt[c][s], sc.nsample[s], lane);
                        if (hit) bb[cnt[c][s] + __builtin_popcountll(hm & lt)] = k;
                        cnt[c][s] += nh;
                        tot[c][s] += nh;
                    }
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int c = 0; c < BC_CPW; ++c) {
        if (cj[c] < 0) continue;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int ns = sc.nsample[s];
            int *bb = buf[wave][c][s];
            const int cc = bc_compact(bb, cnt[c][s], ns, lane);
            int *row = sc.idx[s] + ((size_t)bi * m + cj[c]) * ns;
            // rows are padded with the first (smallest) index, rows without a hit are zero-filled
            const int fill = cc > 0 ? bb[0] : 0;
            for (int l = lane; l < ns; l += 64) row[l] = l < cc ? bb[l] : fill;
            if (sc.pts_cnt[s] && lane == 0) sc.pts_cnt[s][(size_t)bi * m + cj[c]] = cc;
        }
    }
}

static int bq_run(int b, int n, int m, int nscales, const BallScales &sc, const float *xyz1, const float *xyz2,
                  void *ws, size_t ws_bytes, hipStream_t st, const int *order_points = nullptr,
                  const int *order_centroids = nullptr)
{
    // queries = centroids (m), known = points (n)
    NsWs w = ns_carve(ws, b, m, n);
    if (!ws || ws_bytes < w.bytes) return PRCNN_WORKSPACE;
    if (((uintptr_t)ws & 255) != 0) return PRCNN_BAD_ARG;
    if ((order_points == nullptr) != (order_centroids == nullptr)) return PRCNN_BAD_ARG;
    const int tiles = (n + 63) / 64;
    if (order_points) { w.list_q = const_cast<int *>(order_centroids); w.list_k = const_cast<int *>(order_points); }
    else launch_ns_sort(b, m, n, xyz2, xyz1, w.list_q, w.list_k, st);
    const long long total_tiles = (long long)b * tiles;
    hipLaunchKernelGGL(ns_tile_kernel<false>, dim3((unsigned)((total_tiles + 3) / 4)), dim3(256), 0, st, n, tiles,
                       total_tiles, xyz1, w.list_k, w.kp, w.tb);
    dim3 grid((m + 4 * BC_CPW - 1) / (4 * BC_CPW), b);
    switch (nscales) {
        case 1: hipLaunchKernelGGL(bq_culled_kernel<1>, grid, dim3(256), 0, st, n, m, tiles, sc, xyz2, w.list_q, w.kp, w.tb); break;
        case 2: hipLaunchKernelGGL(bq_culled_kernel<2>, grid, dim3(256), 0, st, n, m, tiles, sc, xyz2, w.list_q, w.kp, w.tb); break;
        case 3: hipLaunchKernelGGL(bq_culled_kernel<3>, grid, dim3(256), 0, st, n, m, tiles, sc, xyz2, w.list_q, w.kp, w.tb); break;
        default: hipLaunchKernelGGL(bq_culled_kernel<4>, grid, dim3(256), 0, st, n, m, tiles, sc, xyz2, w.list_q, w.kp, w.tb); break;
    }
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

}  // namespace prcnn

extern "C" size_t prcnn_three_nn_sorted_workspace_bytes(int b, int n, int m)
{
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return prcnn::ns_carve(nullptr, b, n, m).bytes;
}

extern "C" int prcnn_three_nn_sorted(int b, int n, int m, const float *xyz1, const float *xyz2, float *dist,
                                     int *idx, void *workspace, size_t workspace_bytes,
                                     prcnn_stream_t stream)
{
    if (b < 0 || n < 0 || m <= 0) return PRCNN_BAD_SHAPE;   // tf_interpolate.cpp:163,168
    if (b == 0 || n == 0) return PRCNN_OK;
    if (!xyz1 || !xyz2 || !dist || !idx) return PRCNN_BAD_SHAPE;
    if (b > 65535) return PRCNN_UNSUPPORTED;
    return prcnn::ns_run(b, n, m, 0, xyz1, xyz2, nullptr, nullptr, 0, dist, idx, workspace, workspace_bytes,
                         to_stream(stream));
}

extern "C" int prcnn_grid_order(int b, int n, int m, const float *xyz1, const float *xyz2, int *order1, int *order2,
                                prcnn_stream_t stream)
{
    if (b < 0 || n <= 0 || m <= 0) return PRCNN_BAD_SHAPE;
    if (b == 0) return PRCNN_OK;
    if (!xyz1 || !xyz2 || !order1 || !order2) return PRCNN_BAD_SHAPE;
    if (b > 65535) return PRCNN_UNSUPPORTED;
    prcnn::launch_ns_sort(b, n, m, xyz1, xyz2, order1, order2, to_stream(stream));
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_three_nn_ordered(int b, int n, int m, const float *xyz1, const float *xyz2, const int *order1,
                                      const int *order2, float *dist, int *idx, void *workspace,
                                      size_t workspace_bytes, prcnn_stream_t stream)
{
    if (b < 0 || n < 0 || m <= 0) return PRCNN_BAD_SHAPE;
    if (b == 0 || n == 0) return PRCNN_OK;
    if (!xyz1 || !xyz2 || !dist || !idx) return PRCNN_BAD_SHAPE;
    if (b > 65535) return PRCNN_UNSUPPORTED;
    return prcnn::ns_run(b, n, m, 0, xyz1, xyz2, nullptr, nullptr, 0, dist, idx, workspace, workspace_bytes,
                         to_stream(stream), 0, prcnn::EpiAffine{}, order1, order2);
}

extern "C" int prcnn_three_nn_interpolate_sorted(int b, int n, int m, int c, const float *xyz1,
                                                 const float *xyz2, const float *points2, float *out,
                                                 int64_t ldo, float *dist, int *idx, void *workspace,
                                                 size_t workspace_bytes, prcnn_stream_t stream)
{
    if (b < 0 || n < 0 || m <= 0 || c <= 0 || ldo < c) return PRCNN_BAD_SHAPE;
    if (b == 0 || n == 0) return PRCNN_OK;
    if (!xyz1 || !xyz2 || !points2 || !out) return PRCNN_BAD_SHAPE;
    if (b > 65535) return PRCNN_UNSUPPORTED;
    return prcnn::ns_run(b, n, m, c, xyz1, xyz2, points2, out, ldo, dist, idx, workspace, workspace_bytes,
                         to_stream(stream));
}

// culled fused search with an accumulate / ReLU epilogue (entry point in interp.hip)
int prcnn_ns_fused(int b, int n, int m, int c, const float *xyz1, const float *xyz2, const float *points2,
                   float *out, int64_t ldo, float *dist, int *idx, int epi, const prcnn::EpiAffine *aff,
                   void *ws, size_t ws_bytes,
                   prcnn_stream_t stream, const int *order1, const int *order2)
{
    if (b < 0 || n < 0 || m <= 0 || c <= 0 || ldo < c) return PRCNN_BAD_SHAPE;
    if (b == 0 || n == 0) return PRCNN_OK;
    if (!xyz1 || !xyz2 || !points2 || !out) return PRCNN_BAD_SHAPE;
    if (b > 65535) return PRCNN_UNSUPPORTED;
    return prcnn::ns_run(b, n, m, c, xyz1, xyz2, points2, out, ldo, dist, idx, ws, ws_bytes, to_stream(stream), epi,
                         aff ? *aff : prcnn::EpiAffine{}, order1, order2);
}

extern "C" size_t prcnn_query_ball_point_sorted_workspace_bytes(int b, int n, int m)
{
    if (b <= 0 || n <= 0 || m <= 0) return 0;
    return prcnn::ns_carve(nullptr, b, m, n).bytes;
}

extern "C" int prcnn_query_ball_point_msg_ordered(int b, int n, int m, int nscales, const float *host_radii,
                                                  const int *host_nsamples, const float *xyz1,
                                                  const float *xyz2, const int *order1, const int *order2,
                                                  int *const *host_idx, int *const *host_pts_cnt, void *workspace,
                                                  size_t workspace_bytes, prcnn_stream_t stream);

extern "C" int prcnn_query_ball_point_msg_sorted(int b, int n, int m, int nscales, const float *host_radii,
                                                 const int *host_nsamples, const float *xyz1,
                                                 const float *xyz2, int *const *host_idx,
                                                 int *const *host_pts_cnt, void *workspace,
                                                 size_t workspace_bytes, prcnn_stream_t stream)
{
    return prcnn_query_ball_point_msg_ordered(b, n, m, nscales, host_radii, host_nsamples, xyz1, xyz2, nullptr,
                                              nullptr, host_idx, host_pts_cnt, workspace, workspace_bytes, stream);
}

extern "C" int prcnn_query_ball_point_msg_ordered(int b, int n, int m, int nscales, const float *host_radii,
                                                  const int *host_nsamples, const float *xyz1,
                                                  const float *xyz2, const int *order1, const int *order2,
                                                  int *const *host_idx, int *const *host_pts_cnt, void *workspace,
                                                  size_t workspace_bytes, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (nscales < 1 || nscales > BQ_MAX_SCALES) return PRCNN_UNSUPPORTED;
    if (!host_radii || !host_nsamples || !host_idx) return PRCNN_BAD_SHAPE;
    BallScales sc = {};
    for (int s = 0; s < nscales; ++s) {
        if (!(host_radii[s] > 0.0f)) return PRCNN_BAD_ARG;       // tf_grouping.cpp:71
        if (host_nsamples[s] <= 0) return PRCNN_BAD_ARG;         // tf_grouping.cpp:74
        if (host_nsamples[s] > 64) return PRCNN_UNSUPPORTED;     // LDS buffer = 2 x 64 entries
        sc.thr2[s] = ball_threshold(host_radii[s]);
        sc.nsample[s] = host_nsamples[s];
        sc.idx[s] = host_idx[s];
        sc.pts_cnt[s] = host_pts_cnt ? host_pts_cnt[s] : nullptr;
    }
    if (b < 0 || n <= 0 || m < 0) return PRCNN_BAD_SHAPE;
    if (b == 0 || m == 0) return PRCNN_OK;
    if (!xyz1 || !xyz2) return PRCNN_BAD_SHAPE;
    for (int s = 0; s < nscales; ++s)
        if (!sc.idx[s]) return PRCNN_BAD_SHAPE;
    if (b > 65535) return PRCNN_UNSUPPORTED;
    return bq_run(b, n, m, nscales, sc, xyz1, xyz2, workspace, workspace_bytes, to_stream(stream), order1, order2);
}
