// Ball query + group_point for gfx950 (MI355X).
//
// Replaces query_ball_point_gpu / group_point_gpu of the reference
// (models/tf_ops/grouping/tf_grouping_g.cu:3-57; CPU twins query_ball_point.cpp:19-66).
//
// query_ball_point: the reference gives one THREAD a centroid and scans all n points
// serially.  Here one WAVE owns CPW centroids and scans 64 candidate points per step:
// candidate tiles are staged once per workgroup in LDS (SoA, conflict-free), every
// lane tests its point against the wave's centroids, and "first nsample in index
// order" is kept by ballot + prefix-popcount compaction (no atomics, so order is exact).
// The reference's  max(sqrtf(d2),1e-20f) < radius  is evaluated as  d2 < thr2  with
// thr2 = the smallest float whose correctly-rounded sqrt is >= radius (computed on the
// host): identical decisions for every float d2 because sqrtf is monotone, without the
// ~10-instruction IEEE sqrt sequence in the inner loop.  d2 itself uses the reference's
// fp32 expression, left to right, no FMA contraction (-ffp-contract=off).
//
// group_point: pure gather; one lane per 16-byte (or 4-byte) piece of the OUTPUT so
// stores are fully coalesced and each gathered row is read in contiguous pieces.
#include <math.h>
#include "common.h"
#include "ball.h"

namespace prcnn {

constexpr int BQ_THREADS = 256;
constexpr int BQ_WAVES = BQ_THREADS / 64;
constexpr int BQ_CPW = 4;                         // centroids per wave
constexpr int BQ_CPB = BQ_WAVES * BQ_CPW;         // centroids per workgroup
constexpr int BQ_TILE = 2048;                     // candidate points per LDS tile


template <int NS>
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(int n, int m, const BallScales sc,
                                                                const float *__restrict__ xyz1,
                                                                const float *__restrict__ xyz2)
{
    // the tile is a straight copy of the cloud (x,y,z interleaved): a lane reads its point at
    // word 3*lane + {0,1,2}; stride 3 is coprime with the 64 banks, so the reads are conflict-free
    __shared__ __attribute__((aligned(16))) float sp[BQ_TILE * 3];
    const int tid = threadIdx.x, lane = tid & 63;
    // wave-uniform by construction; readfirstlane tells the compiler, so that the per-centroid
    // counters, flags and centroid coordinates live in SGPRs and all control flow is scalar
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bi = blockIdx.y;
    const float *pts = xyz1 + (size_t)bi * n * 3;
    const float *cen = xyz2 + (size_t)bi * m * 3;

    const int j0 = (blockIdx.x * BQ_WAVES + wave) * BQ_CPW;
    float cx[BQ_CPW], cy[BQ_CPW], cz[BQ_CPW];
    int cnt[BQ_CPW][NS], first[BQ_CPW][NS];
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) {
        const int j = j0 + c;
        const bool ok = j < m;
        cx[c] = ok ? cen[j * 3 + 0] : 0.f;
        cy[c] = ok ? cen[j * 3 + 1] : 0.f;
        cz[c] = ok ? cen[j * 3 + 2] : 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            cnt[c][s] = ok ? 0 : sc.nsample[s];          // out-of-range centroid: already "full"
            first[c][s] = 0;
        }
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    // thr_open[c]: largest threshold among the scales of centroid c that still take points
    // (-inf once all are full), so ONE compare per centroid screens a 64-point step and the
    // per-scale bookkeeping below only runs for steps that contain a hit.
    float thr_open[BQ_CPW];
    auto refresh = [&](int c) {
        float t = -INFINITY;
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (cnt[c][s] < sc.nsample[s]) t = fmaxf(t, sc.thr2[s]);
        thr_open[c] = t;
    };
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) refresh(c);

    const float nanv = __builtin_nanf("");
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (int base = 0; base < n; base += BQ_TILE) {
        const int tn = min(BQ_TILE, n - base);
        const int tn64 = (tn + 63) & ~63;
        __syncthreads();
        {
            const float *src = pts + (size_t)base * 3;
            const int nf = tn * 3;
            if ((((uintptr_t)src) & 15) == 0) {
                const int nv = nf >> 2;
                for (int e = tid; e < nv; e += BQ_THREADS)
                    reinterpret_cast<float4 *>(sp)[e] = reinterpret_cast<const float4 *>(src)[e];
                for (int e = (nv << 2) + tid; e < nf; e += BQ_THREADS) sp[e] = src[e];
            } else {
                for (int e = tid; e < nf; e += BQ_THREADS) sp[e] = src[e];
            }
        }
        // tail of the last 64-point chunk: NaN coordinates never pass  d2 < thr2
        for (int e = tn * 3 + tid; e < tn64 * 3; e += BQ_THREADS) sp[e] = nanv;
        __syncthreads();
        bool open = false;                           // wave-uniform: some centroid still takes points
#pragma unroll
        for (int c = 0; c < BQ_CPW; ++c) open = open || thr_open[c] > -INFINITY;
        for (int p0 = 0; p0 < tn64 && open; p0 += 64) {
            const float *pp = sp + (p0 + lane) * 3;
            const float x1 = pp[0], y1 = pp[1], z1 = pp[2];
            const int k = base + p0 + lane;
            // two centroids per packed-fp32 operation (v_pk_add/mul_f32: each component is the
            // same IEEE operation as the scalar form, so decisions are bit-identical)
            const f32x2 px = {x1, x1}, py = {y1, y1}, pz = {z1, z1};
            float d2[BQ_CPW];
            unsigned long long any = 0;
#pragma unroll
            for (int c0 = 0; c0 < BQ_CPW; c0 += 2) {
                const f32x2 ccx = {cx[c0], cx[c0 + 1]}, ccy = {cy[c0], cy[c0 + 1]}, ccz = {cz[c0], cz[c0 + 1]};
                const f32x2 dx = ccx - px, dy = ccy - py, dz = ccz - pz;
                const f32x2 d2p = dx * dx + dy * dy + dz * dz;
                d2[c0] = d2p[0]; d2[c0 + 1] = d2p[1];
                any |= __builtin_amdgcn_ballot_w64(d2p[0] < thr_open[c0]);
                any |= __builtin_amdgcn_ballot_w64(d2p[1] < thr_open[c0 + 1]);
            }
            if (any == 0) continue;
#pragma unroll
            for (int c = 0; c < BQ_CPW; ++c) {
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (cnt[c][s] < sc.nsample[s]) {
                        const bool hit = d2[c] < sc.thr2[s];
                        const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                        if (mask) {
                            if (cnt[c][s] == 0) first[c][s] = base + p0 + __builtin_ctzll(mask);
                            const int slot = cnt[c][s] + __builtin_popcountll(mask & lt);
                            if (hit && slot < sc.nsample[s])
                                sc.idx[s][((size_t)bi * m + j0 + c) * sc.nsample[s] + slot] = k;
                            cnt[c][s] += __builtin_popcountll(mask);
                        }
                    }
                }
                refresh(c);
            }
            open = false;
#pragma unroll
            for (int c = 0; c < BQ_CPW; ++c) open = open || thr_open[c] > -INFINITY;
        }
        // the whole workgroup leaves once every centroid of every wave is full
        if (__syncthreads_and(!open)) break;
    }
    // pad rows that found fewer than nsample points with the first hit
    // (tf_grouping_g.cu:26-29); rows without any hit are zero-filled.
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) {
        const int j = j0 + c;
        if (j < m) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int ns = sc.nsample[s];
                const int cc = min(cnt[c][s], ns);
                int *row = sc.idx[s] + ((size_t)bi * m + j) * ns;
                // no hit at all: the reference leaves the row uninitialised; write zeros so the
                // result is deterministic (and callers need not pre-clear idx)
                const int fill = cc > 0 ? first[c][s] : 0;
                for (int l = cc + lane; l < ns; l += 64) row[l] = fill;
                if (sc.pts_cnt[s] && lane == 0) sc.pts_cnt[s][(size_t)bi * m + j] = cc;
            }
        }
    }
}

// out[row, :] = points[b, idx[row], :]; one lane per VEC floats of the output.
template <int VEC>
__global__ __launch_bounds__(256) void group_point_kernel(int n, int c, long long rows_per_batch,
                                                          long long total_vec,
                                                          const float *__restrict__ points,
                                                          const int *__restrict__ idx,
                                                          float *__restrict__ out)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int cv = c / VEC;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total_vec;
         e += (long long)gridDim.x * blockDim.x) {
        const long long row = e / cv;
        const int col = (int)(e - row * cv);
        const long long bi = row / rows_per_batch;
        const int ii = idx[row];
        const vec_t v = *reinterpret_cast<const vec_t *>(points + ((size_t)bi * n + ii) * c + col * VEC);
        *reinterpret_cast<vec_t *>(out + (size_t)e * VEC) = v;
    }
}

// smallest float t such that max(sqrtf(t), 1e-20f) >= radius, i.e. the reference's
// test  max(sqrtf(d2),1e-20f) < radius  <=>  d2 < t   for every float d2 >= 0.
float ball_threshold(float radius)
{
    if (!(radius > 1e-20f)) return 0.0f;            // d >= 1e-20 >= radius: nothing passes
    float t = (float)((double)radius * (double)radius);
    if (isinf(t)) t = 3.402823466e38f;
    while (t > 0.0f && sqrtf(t) >= radius) t = nextafterf(t, 0.0f);
    while (sqrtf(t) < radius) {
        if (isinf(t)) break;
        t = nextafterf(t, INFINITY);
    }
    return t;
}

}  // namespace prcnn

extern "C" int prcnn_query_ball_point_msg(int b, int n, int m, int nscales, const float *host_radii,
                                          const int *host_nsamples, const float *xyz1,
                                          const float *xyz2, int *const *host_idx,
                                          int *const *host_pts_cnt, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (nscales < 1 || nscales > BQ_MAX_SCALES) return PRCNN_UNSUPPORTED;
    if (!host_radii || !host_nsamples || !host_idx) return PRCNN_BAD_SHAPE;
    BallScales sc = {};
    for (int s = 0; s < nscales; ++s) {
        if (!(host_radii[s] > 0.0f)) return PRCNN_BAD_ARG;       // tf_grouping.cpp:71
        if (host_nsamples[s] <= 0) return PRCNN_BAD_ARG;         // tf_grouping.cpp:74
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
    dim3 grid((m + BQ_CPB - 1) / BQ_CPB, b);
    hipStream_t st = to_stream(stream);
    switch (nscales) {
        case 1: hipLaunchKernelGGL(ball_query_kernel<1>, grid, dim3(BQ_THREADS), 0, st, n, m, sc, xyz1, xyz2); break;
        case 2: hipLaunchKernelGGL(ball_query_kernel<2>, grid, dim3(BQ_THREADS), 0, st, n, m, sc, xyz1, xyz2); break;
        case 3: hipLaunchKernelGGL(ball_query_kernel<3>, grid, dim3(BQ_THREADS), 0, st, n, m, sc, xyz1, xyz2); break;
        default: hipLaunchKernelGGL(ball_query_kernel<4>, grid, dim3(BQ_THREADS), 0, st, n, m, sc, xyz1, xyz2); break;
    }
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_query_ball_point(int b, int n, int m, float radius, int nsample,
                                      const float *xyz1, const float *xyz2, int *idx,
                                      int *pts_cnt, prcnn_stream_t stream)
{
    if (!idx && b > 0 && m > 0 && radius > 0.0f && nsample > 0 && n > 0) return PRCNN_BAD_SHAPE;
    int *idxs[1] = {idx};
    int *cnts[1] = {pts_cnt};
    return prcnn_query_ball_point_msg(b, n, m, 1, &radius, &nsample, xyz1, xyz2, idxs, cnts, stream);
}

extern "C" int prcnn_group_point(int b, int n, int c, int m, int nsample, const float *points,
                                 const int *idx, float *out, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (b < 0 || n <= 0 || c < 0 || m < 0 || nsample < 0) return PRCNN_BAD_SHAPE;
    const long long rows = (long long)b * m * nsample;
    if (rows == 0 || c == 0) return PRCNN_OK;
    if (!points || !idx || !out) return PRCNN_BAD_SHAPE;
    const long long rpb = (long long)m * nsample;
    hipStream_t st = to_stream(stream);
    const bool a16 = (((uintptr_t)points | (uintptr_t)out) & 15) == 0;
    if (c % 4 == 0 && a16) {
        const long long tv = rows * (c / 4);
        long long blocks = (tv + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(group_point_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, n, c, rpb,
                           tv, points, idx, out);
    } else {
        const long long tv = rows * c;
        long long blocks = (tv + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(group_point_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, n, c, rpb,
                           tv, points, idx, out);
    }
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}
