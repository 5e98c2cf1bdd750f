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

namespace prcnn {

constexpr int BQ_THREADS = 256;
constexpr int BQ_WAVES = BQ_THREADS / 64;
constexpr int BQ_CPW = 4;                         // centroids per wave
constexpr int BQ_CPB = BQ_WAVES * BQ_CPW;         // centroids per workgroup
constexpr int BQ_TILE = 2048;                     // candidate points per LDS tile

__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(
    int n, int m, float thr2, int nsample, const float *__restrict__ xyz1,
    const float *__restrict__ xyz2, int *__restrict__ idx, int *__restrict__ pts_cnt)
{
    __shared__ float sx[BQ_TILE], sy[BQ_TILE], sz[BQ_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bi = blockIdx.y;
    const float *pts = xyz1 + (size_t)bi * n * 3;
    const float *cen = xyz2 + (size_t)bi * m * 3;
    int *oidx = idx + (size_t)bi * m * nsample;

    const int j0 = (blockIdx.x * BQ_WAVES + wave) * BQ_CPW;
    float cx[BQ_CPW], cy[BQ_CPW], cz[BQ_CPW];
    int cnt[BQ_CPW], first[BQ_CPW];
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) {
        const int j = j0 + c;
        const bool ok = j < m;
        cx[c] = ok ? cen[j * 3 + 0] : 0.f;
        cy[c] = ok ? cen[j * 3 + 1] : 0.f;
        cz[c] = ok ? cen[j * 3 + 2] : 0.f;
        cnt[c] = ok ? 0 : nsample;                  // out-of-range centroid: already "full"
        first[c] = 0;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;

    for (int base = 0; base < n; base += BQ_TILE) {
        const int tn = min(BQ_TILE, n - base);
        __syncthreads();
        for (int e = tid; e < tn * 3; e += BQ_THREADS) {
            const float v = pts[(size_t)base * 3 + e];
            const int p = e / 3, c = e - p * 3;
            (c == 0 ? sx : c == 1 ? sy : sz)[p] = v;
        }
        __syncthreads();
        bool wave_done = true;
#pragma unroll
        for (int c = 0; c < BQ_CPW; ++c) wave_done = wave_done && cnt[c] >= nsample;
        if (!wave_done) {
            for (int p0 = 0; p0 < tn; p0 += 64) {
                const int p = p0 + lane;
                const bool valid = p < tn;
                const float x1 = valid ? sx[p] : 0.f;
                const float y1 = valid ? sy[p] : 0.f;
                const float z1 = valid ? sz[p] : 0.f;
                const int k = base + p;
                bool all_done = true;
#pragma unroll
                for (int c = 0; c < BQ_CPW; ++c) {
                    if (cnt[c] < nsample) {          // wave-uniform
                        const float dx = cx[c] - x1, dy = cy[c] - y1, dz = cz[c] - z1;
                        const float d2 = dx * dx + dy * dy + dz * dz;
                        const bool hit = valid && d2 < thr2;
                        const unsigned long long mask = __ballot(hit);
                        if (mask) {
                            if (cnt[c] == 0) first[c] = base + p0 + __builtin_ctzll(mask);
                            const int slot = cnt[c] + __builtin_popcountll(mask & lt);
                            if (hit && slot < nsample) oidx[(size_t)(j0 + c) * nsample + slot] = k;
                            cnt[c] += __builtin_popcountll(mask);
                        }
                        all_done = all_done && cnt[c] >= nsample;
                    }
                }
                if (all_done) break;
            }
        }
        // the whole workgroup leaves once every centroid of every wave is full
        bool done = true;
#pragma unroll
        for (int c = 0; c < BQ_CPW; ++c) done = done && cnt[c] >= nsample;
        if (__syncthreads_and(done)) break;
    }
    // pad rows that found fewer than nsample points with the first hit
    // (tf_grouping_g.cu:26-29); rows without any hit stay unwritten.
#pragma unroll
    for (int c = 0; c < BQ_CPW; ++c) {
        const int j = j0 + c;
        if (j < m) {
            const int cc = min(cnt[c], nsample);
            if (cc > 0)
                for (int l = cc + lane; l < nsample; l += 64) oidx[(size_t)j * nsample + l] = first[c];
            if (pts_cnt && lane == 0) pts_cnt[(size_t)bi * m + j] = cc;
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
static float ball_threshold(float radius)
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

extern "C" int prcnn_query_ball_point(int b, int n, int m, float radius, int nsample,
                                      const float *xyz1, const float *xyz2, int *idx,
                                      int *pts_cnt, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (!(radius > 0.0f)) return PRCNN_BAD_ARG;       // tf_grouping.cpp:71
    if (nsample <= 0) return PRCNN_BAD_ARG;           // tf_grouping.cpp:74
    if (b < 0 || n <= 0 || m < 0) return PRCNN_BAD_SHAPE;
    if (b == 0 || m == 0) return PRCNN_OK;
    if (!xyz1 || !xyz2 || !idx) return PRCNN_BAD_SHAPE;
    if (b > 65535) return PRCNN_UNSUPPORTED;
    const float thr2 = ball_threshold(radius);
    dim3 grid((m + BQ_CPB - 1) / BQ_CPB, b);
    hipLaunchKernelGGL(ball_query_kernel, grid, dim3(BQ_THREADS), 0, to_stream(stream), n, m, thr2,
                       nsample, xyz1, xyz2, idx, pts_cnt);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
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
