// three_nn / three_interpolate / FP weights and their fusion for gfx950 (MI355X).
//
// Replaces threenn_cpu / threeinterpolate_cpu of the reference
// (models/tf_ops/3d_interpolation/tf_interpolate.cpp:60-127, registered DEVICE_CPU only,
// so every FP layer of the reference round-trips through the host) and the TF
// elementwise weight computation of models/pointnet_util.py:211-214.
//
// three_nn: one lane per query point; the known points are staged in LDS as float4 and
// read with a wave-uniform address (broadcast ds_read_b128, one per test).  Squared fp32
// distance in the reference's left-to-right order (no FMA), 3-slot strict-'<' insertion
// (ties keep the earlier index).  The reference holds `best` in a double initialised to
// 1e40: for fp32 d that is indistinguishable from +inf, which is what is used here and
// what the reference's float output becomes when m < 3.
// three_interpolate: one lane per 16 B of output, ((p1*w1)+(p2*w2))+(p3*w3) unfused.
// Fused kernel: phase 1 = three_nn + weights for the workgroup's queries into LDS,
// phase 2 = the workgroup sweeps its output rows channel-contiguously.
#include "common.h"

namespace prcnn {

constexpr int NN_THREADS = 256;
constexpr int NN_TILE = 2048;

struct Top3 {
    float d1, d2, d3;
    int i1, i2, i3;
};

__device__ __forceinline__ void top3_insert(Top3 &t, float d, int kk)
{
    if (d < t.d3) {                                   // tf_interpolate.cpp:74-89
        if (d < t.d1) {
            t.d3 = t.d2; t.i3 = t.i2;
            t.d2 = t.d1; t.i2 = t.i1;
            t.d1 = d; t.i1 = kk;
        } else if (d < t.d2) {
            t.d3 = t.d2; t.i3 = t.i2;
            t.d2 = d; t.i2 = kk;
        } else {
            t.d3 = d; t.i3 = kk;
        }
    }
}

__device__ __forceinline__ float sqdist(const float4 p, float x1, float y1, float z1)
{
    const float dx = p.x - x1, dy = p.y - y1, dz = p.z - z1;
    return dx * dx + dy * dy + dz * dz;               // tf_interpolate.cpp:73, unfused
}

// The tile is padded to a multiple of 4 with NaN points (never smaller than anything).
// Four candidates per step; the insertion chain only runs when one of them can enter the
// top 3, and then visits them in index order, so ties resolve exactly as in the reference.
__device__ __forceinline__ void top3_scan(Top3 &t, float x1, float y1, float z1,
                                          const float4 *__restrict__ tile, int tn, int base)
{
    const int tn4 = (tn + 3) & ~3;
    for (int k = 0; k < tn4; k += 4) {
        const float d0 = sqdist(tile[k], x1, y1, z1);
        const float d1 = sqdist(tile[k + 1], x1, y1, z1);
        const float d2 = sqdist(tile[k + 2], x1, y1, z1);
        const float d3 = sqdist(tile[k + 3], x1, y1, z1);
        const float mn = fminf(fminf(d0, d1), fminf(d2, d3));
        if (mn < t.d3) {
            top3_insert(t, d0, base + k);
            top3_insert(t, d1, base + k + 1);
            top3_insert(t, d2, base + k + 2);
            top3_insert(t, d3, base + k + 3);
        }
    }
}

__device__ __forceinline__ void stage_tile(float4 *tile, const float *__restrict__ known, int base,
                                           int tn, int tid, int nthreads)
{
    const int tn4 = (tn + 3) & ~3;
    const float nan = __builtin_nanf("");
    for (int k = tid; k < tn4; k += nthreads) {
        if (k < tn) {
            const float *p = known + (size_t)(base + k) * 3;
            tile[k] = make_float4(p[0], p[1], p[2], 0.f);
        } else {
            tile[k] = make_float4(nan, nan, nan, 0.f);
        }
    }
}

__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(int n, int m,
                                                              const float *__restrict__ xyz1,
                                                              const float *__restrict__ xyz2,
                                                              float *__restrict__ dist,
                                                              int *__restrict__ idx)
{
    __shared__ float4 tile[NN_TILE];
    const int bi = blockIdx.y;
    const int j = blockIdx.x * NN_THREADS + threadIdx.x;
    const bool ok = j < n;
    const float *q = xyz1 + ((size_t)bi * n + (ok ? j : 0)) * 3;
    const float x1 = q[0], y1 = q[1], z1 = q[2];
    const float *known = xyz2 + (size_t)bi * m * 3;
    Top3 t = {INFINITY, INFINITY, INFINITY, 0, 0, 0};
    for (int base = 0; base < m; base += NN_TILE) {
        const int tn = min(NN_TILE, m - base);
        __syncthreads();
        stage_tile(tile, known, base, tn, threadIdx.x, NN_THREADS);
        __syncthreads();
        top3_scan(t, x1, y1, z1, tile, tn, base);
    }
    if (ok) {
        const size_t o = ((size_t)bi * n + j) * 3;
        dist[o + 0] = t.d1; dist[o + 1] = t.d2; dist[o + 2] = t.d3;
        idx[o + 0] = t.i1; idx[o + 1] = t.i2; idx[o + 2] = t.i3;
    }
}

// pointnet_util.py:211-214
__device__ __forceinline__ void fp_weights3(float d0, float d1, float d2, float &w0, float &w1,
                                            float &w2)
{
    d0 = fmaxf(d0, 1e-10f); d1 = fmaxf(d1, 1e-10f); d2 = fmaxf(d2, 1e-10f);
    const float r0 = 1.0f / d0, r1 = 1.0f / d1, r2 = 1.0f / d2;
    const float norm = (r0 + r1) + r2;
    w0 = r0 / norm; w1 = r1 / norm; w2 = r2 / norm;
}

__global__ void fp_weights_kernel(long long rows, const float *__restrict__ dist,
                                  float *__restrict__ weight)
{
    for (long long r = blockIdx.x * (long long)blockDim.x + threadIdx.x; r < rows;
         r += (long long)gridDim.x * blockDim.x) {
        float w0, w1, w2;
        fp_weights3(dist[r * 3], dist[r * 3 + 1], dist[r * 3 + 2], w0, w1, w2);
        weight[r * 3] = w0; weight[r * 3 + 1] = w1; weight[r * 3 + 2] = w2;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void three_interpolate_kernel(int m, int c, int n, long long total_vec,
                                                                const float *__restrict__ points,
                                                                const int *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ out)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int cv = c / VEC;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total_vec;
         e += (long long)gridDim.x * blockDim.x) {
        const long long row = e / cv;                 // row = b*n + j
        const int col = (int)(e - row * cv) * VEC;
        const long long bi = row / n;
        const float w1 = weight[row * 3], w2 = weight[row * 3 + 1], w3 = weight[row * 3 + 2];
        const int i1 = idx[row * 3], i2 = idx[row * 3 + 1], i3 = idx[row * 3 + 2];
        const float *pb = points + (size_t)bi * m * c + col;
        const vec_t a = *reinterpret_cast<const vec_t *>(pb + (size_t)i1 * c);
        const vec_t bb = *reinterpret_cast<const vec_t *>(pb + (size_t)i2 * c);
        const vec_t cc = *reinterpret_cast<const vec_t *>(pb + (size_t)i3 * c);
        const vec_t r = a * w1 + bb * w2 + cc * w3;   // unfused: -ffp-contract=off
        *reinterpret_cast<vec_t *>(out + (size_t)row * c + col) = r;
    }
}

constexpr int FU_THREADS = 256;

template <int VEC>
__global__ __launch_bounds__(FU_THREADS) void three_nn_interpolate_kernel(
    int n, int m, int c, const float *__restrict__ xyz1, const float *__restrict__ xyz2,
    const float *__restrict__ points2, float *__restrict__ out, long long ldo,
    float *__restrict__ dist, int *__restrict__ idx)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ float4 tile[NN_TILE];
    __shared__ int s_i[FU_THREADS * 3];
    __shared__ float s_w[FU_THREADS * 3];
    const int tid = threadIdx.x;
    const int bi = blockIdx.y;
    const int j0 = blockIdx.x * FU_THREADS;
    const int j = j0 + tid;
    const bool ok = j < n;
    const float *q = xyz1 + ((size_t)bi * n + (ok ? j : 0)) * 3;
    const float x1 = q[0], y1 = q[1], z1 = q[2];
    const float *known = xyz2 + (size_t)bi * m * 3;
    Top3 t = {INFINITY, INFINITY, INFINITY, 0, 0, 0};
    for (int base = 0; base < m; base += NN_TILE) {
        const int tn = min(NN_TILE, m - base);
        __syncthreads();
        stage_tile(tile, known, base, tn, tid, FU_THREADS);
        __syncthreads();
        top3_scan(t, x1, y1, z1, tile, tn, base);
    }
    float w0, w1, w2;
    fp_weights3(t.d1, t.d2, t.d3, w0, w1, w2);
    s_i[tid * 3] = t.i1; s_i[tid * 3 + 1] = t.i2; s_i[tid * 3 + 2] = t.i3;
    s_w[tid * 3] = w0; s_w[tid * 3 + 1] = w1; s_w[tid * 3 + 2] = w2;
    if (ok) {
        const size_t o = ((size_t)bi * n + j) * 3;
        if (dist) { dist[o] = t.d1; dist[o + 1] = t.d2; dist[o + 2] = t.d3; }
        if (idx) { idx[o] = t.i1; idx[o + 1] = t.i2; idx[o + 2] = t.i3; }
    }
    __syncthreads();
    const int rows = min(FU_THREADS, n - j0);
    const int cv = c / VEC;
    const float *pb = points2 + (size_t)bi * m * c;
    for (int e = tid; e < rows * cv; e += FU_THREADS) {
        const int r = e / cv;
        const int col = (e - r * cv) * VEC;
        const vec_t a = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3] * c + col);
        const vec_t bb = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3 + 1] * c + col);
        const vec_t cc = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3 + 2] * c + col);
        const vec_t v = a * s_w[r * 3] + bb * s_w[r * 3 + 1] + cc * s_w[r * 3 + 2];
        *reinterpret_cast<vec_t *>(out + ((size_t)bi * n + j0 + r) * ldo + col) = v;
    }
}

}  // namespace prcnn

extern "C" int prcnn_three_nn(int b, int n, int m, const float *xyz1, const float *xyz2,
                              float *dist, int *idx, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (b < 0 || n < 0 || m <= 0) return PRCNN_BAD_SHAPE;   // tf_interpolate.cpp:163,168
    if (b == 0 || n == 0) return PRCNN_OK;
    if (!xyz1 || !xyz2 || !dist || !idx) return PRCNN_BAD_SHAPE;
    if (b > 65535) return PRCNN_UNSUPPORTED;
    dim3 grid((n + NN_THREADS - 1) / NN_THREADS, b);
    hipLaunchKernelGGL(three_nn_kernel, grid, dim3(NN_THREADS), 0, to_stream(stream), n, m, xyz1,
                       xyz2, dist, idx);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_fp_weights(int64_t rows, const float *dist, float *weight,
                                prcnn_stream_t stream)
{
    if (rows < 0) return PRCNN_BAD_SHAPE;
    if (rows == 0) return PRCNN_OK;
    if (!dist || !weight) return PRCNN_BAD_SHAPE;
    long long blocks = (rows + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(prcnn::fp_weights_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       to_stream(stream), (long long)rows, dist, weight);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_three_interpolate(int b, int m, int c, int n, const float *points,
                                       const int *idx, const float *weight, float *out,
                                       prcnn_stream_t stream)
{
    using namespace prcnn;
    if (b < 0 || m <= 0 || c < 0 || n < 0) return PRCNN_BAD_SHAPE;   // tf_interpolate.cpp:203,206
    const long long rows = (long long)b * n;
    if (rows == 0 || c == 0) return PRCNN_OK;
    if (!points || !idx || !weight || !out) return PRCNN_BAD_SHAPE;
    hipStream_t st = to_stream(stream);
    const bool a16 = (((uintptr_t)points | (uintptr_t)out) & 15) == 0;
    if (c % 4 == 0 && a16) {
        const long long tv = rows * (c / 4);
        long long blocks = (tv + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(three_interpolate_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, m, c,
                           n, tv, points, idx, weight, out);
    } else {
        const long long tv = rows * c;
        long long blocks = (tv + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(three_interpolate_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, m, c,
                           n, tv, points, idx, weight, out);
    }
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_three_nn_interpolate(int b, int n, int m, int c, const float *xyz1,
                                          const float *xyz2, const float *points2, float *out,
                                          int64_t ldo, float *dist, int *idx, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (b < 0 || n < 0 || m <= 0 || c <= 0 || ldo < c) return PRCNN_BAD_SHAPE;
    if (b == 0 || n == 0) return PRCNN_OK;
    if (!xyz1 || !xyz2 || !points2 || !out) return PRCNN_BAD_SHAPE;
    if (b > 65535) return PRCNN_UNSUPPORTED;
    hipStream_t st = to_stream(stream);
    dim3 grid((n + FU_THREADS - 1) / FU_THREADS, b);
    const bool a16 = (((uintptr_t)points2 | (uintptr_t)out) & 15) == 0 && ldo % 4 == 0;
    if (c % 4 == 0 && a16)
        hipLaunchKernelGGL(three_nn_interpolate_kernel<4>, grid, dim3(FU_THREADS), 0, st, n, m, c, xyz1,
                           xyz2, points2, out, (long long)ldo, dist, idx);
    else
        hipLaunchKernelGGL(three_nn_interpolate_kernel<1>, grid, dim3(FU_THREADS), 0, st, n, m, c, xyz1,
                           xyz2, points2, out, (long long)ldo, dist, idx);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}
