// three_nn (+ fused interpolation) with spatial culling, for large n x m.
//
// Same contract and bit-identical results as prcnn_three_nn / prcnn_three_nn_interpolate
// (reference: threenn_cpu, models/tf_ops/3d_interpolation/tf_interpolate.cpp:60-103), but
// the b*n*m brute-force tests are cut by an order of magnitude:
//   1. per cloud, queries and known points are bucketed into a uniform Morton-ordered grid
//      over their joint bounding box (integer-atomic counting sort, csr.h);
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
#include "csr.h"

namespace prcnn {

constexpr int NS_THREADS = 256;
constexpr int NS_CHUNK = 2048;            // known points per LDS pass
constexpr int NS_BITS = 5;                // grid cells per axis = 32 (uniform cell size)
constexpr int NS_CELLS = 1 << (3 * NS_BITS);

// ---- 1. joint bounding box per cloud -----------------------------------------------------
__global__ __launch_bounds__(1024) void ns_bbox_kernel(int n, int m, const float *__restrict__ xyz1,
                                                       const float *__restrict__ xyz2,
                                                       float *__restrict__ bbox)
{
    __shared__ float red[16][6];
    const int bi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const float *a = xyz1 + (size_t)bi * n * 3;
    const float *k = xyz2 + (size_t)bi * m * 3;
    for (int i = tid; i < n + m; i += 1024) {
        const float *p = i < n ? a + (size_t)i * 3 : k + (size_t)(i - n) * 3;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = p[d];
            lo[d] = fminf(lo[d], v);          // NaN coordinates are ignored by fmin/fmax
            hi[d] = fmaxf(hi[d], v);
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
    __syncthreads();
    if (tid < 6) {
        float v = red[0][tid];
        for (int w = 1; w < 16; ++w) v = tid < 3 ? fminf(v, red[w][tid]) : fmaxf(v, red[w][tid]);
        bbox[bi * 8 + tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float ext = 0.f;
        for (int d = 0; d < 3; ++d) ext = fmaxf(ext, bbox[bi * 8 + 3 + d] - bbox[bi * 8 + d]);
        // cells per unit length; 0 for a degenerate or non-finite cloud (everything in cell 0)
        const float inv = (ext > 0.f && ext < INFINITY) ? (float)(1 << NS_BITS) / ext : 0.f;
        bbox[bi * 8 + 6] = inv;
    }
}

__device__ __forceinline__ unsigned spread3(unsigned v)    // 5 bits -> every third bit
{
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x30000ffu;
    v = (v | (v << 8)) & 0x300f00fu;
    v = (v | (v << 4)) & 0x30c30c3u;
    v = (v | (v << 2)) & 0x9249249u;
    return v;
}

// ---- 2. Morton cell of every point -------------------------------------------------------
__global__ void ns_key_kernel(long long total, int npts, const float *__restrict__ xyz,
                              const float *__restrict__ bbox, int *__restrict__ keys)
{
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long bi = g / npts;
        const float *bb = bbox + bi * 8;
        const float inv = bb[6];
        unsigned c[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float f = (xyz[g * 3 + d] - bb[d]) * inv;
            int q = f > 0.f ? (int)fminf(f, (float)((1 << NS_BITS) - 1)) : 0;   // NaN -> 0
            c[d] = (unsigned)q;
        }
        keys[g] = (int)(spread3(c[0]) | (spread3(c[1]) << 1) | (spread3(c[2]) << 2));
    }
}

// ---- 3. sorted known points (x, y, z, original index) and their tile boxes -----------------
// one wave per tile of 64 sorted known points; tiles never straddle a chunk of NS_CHUNK
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
    kp[bi * tiles_per_cloud * 64 + s] = p;
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

__device__ __forceinline__ float sqd(const float4 p, float x1, float y1, float z1)
{
    const float dx = p.x - x1, dy = p.y - y1, dz = p.z - z1;
    return dx * dx + dy * dy + dz * dz;               // tf_interpolate.cpp:73, unfused
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
template <int VEC, bool INTERP>
__global__ __launch_bounds__(NS_THREADS) void ns_search_kernel(
    int n, int m, int c, int tiles_per_cloud, const float *__restrict__ xyz1,
    const int *__restrict__ qsorted,            // (b, n) query indices in Morton order
    const float4 *__restrict__ kp, const float *__restrict__ tb, const float *__restrict__ points2,
    float *__restrict__ out, long long ldo, float *__restrict__ dist, int *__restrict__ idx)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    __shared__ float4 tile[NS_CHUNK];
    __shared__ int s_q[NS_THREADS];
    __shared__ int s_i[NS_THREADS * 3];
    __shared__ float s_w[NS_THREADS * 3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int bi = blockIdx.y;
    const int j0 = blockIdx.x * NS_THREADS;
    const int js = j0 + tid;                       // position in sorted order
    const bool ok = js < n;
    const int qi = qsorted[(size_t)bi * n + (ok ? js : n - 1)];
    const float *q = xyz1 + ((size_t)bi * n + qi) * 3;
    const float x1 = q[0], y1 = q[1], z1 = q[2];
    // the wave's query box (lanes past n repeat the last query: no effect on the box)
    const float qlo0 = wave_minf(x1), qlo1 = wave_minf(y1), qlo2 = wave_minf(z1);
    const float qhi0 = wave_maxf(x1), qhi1 = wave_maxf(y1), qhi2 = wave_maxf(z1);
    const float4 *kpc = kp + (size_t)bi * tiles_per_cloud * 64;
    const float *tbc = tb + (size_t)bi * tiles_per_cloud * 8;
    Top3K t = top3k_init();
    const int mpad = tiles_per_cloud * 64;
    for (int base = 0; base < mpad; base += NS_CHUNK) {
        const int tn = min(NS_CHUNK, mpad - base);           // multiple of 64
        __syncthreads();
        for (int k = tid; k < tn; k += NS_THREADS) tile[k] = kpc[base + k];
        __syncthreads();
        // lane l: lower bound between the wave's query box and tile l of this chunk
        const int nt = tn / 64;
        float lb = INFINITY;
        if (lane < nt) {
            const float *bx = tbc + (size_t)(base / 64 + lane) * 8;
            const float g0 = fmaxf(fmaxf(bx[0] - qhi0, qlo0 - bx[3]), 0.f);
            const float g1 = fmaxf(fmaxf(bx[1] - qhi1, qlo1 - bx[4]), 0.f);
            const float g2 = fmaxf(fmaxf(bx[2] - qhi2, qlo2 - bx[5]), 0.f);
            lb = g0 * g0 + g1 * g1 + g2 * g2;                // same op order as sqd(): unfused
            if (!(lb == lb)) lb = 0.f;                        // NaN box: never skip
        }
        int lbi = lane < nt ? f2i(lb) : 0x7fffffff;           // lb >= 0: int order = float order
        for (;;) {
            // nearest unvisited tile of the chunk
            int mn = lbi;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mn = min(mn, __shfl_xor(mn, o));
            if (mn == 0x7fffffff) break;
            const float worst = wave_maxf(ok ? t.d3 : 0.f);
            if (i2f(mn) > worst) break;                       // every remaining tile is farther
            const unsigned long long who = __ballot(lbi == mn);
            const int tl = __builtin_ctzll(who);
            if (lane == tl) lbi = 0x7fffffff;
            const float4 *tp = tile + tl * 64;
#pragma unroll 4
            for (int k = 0; k < 64; k += 4) {
                const float4 p0 = tp[k], p1 = tp[k + 1], p2 = tp[k + 2], p3 = tp[k + 3];
                const float d0 = sqd(p0, x1, y1, z1), d1 = sqd(p1, x1, y1, z1);
                const float d2 = sqd(p2, x1, y1, z1), d3 = sqd(p3, x1, y1, z1);
                const float mn4 = fminf(fminf(d0, d1), fminf(d2, d3));
                // '<=': tiles are not visited in index order, an equal distance with a smaller
                // index must still get in (the 64-bit key decides)
                if (__builtin_amdgcn_ballot_w64(mn4 <= t.d3)) {
                    if (__builtin_amdgcn_ballot_w64(d0 <= t.d3)) top3k_insert(t, d0, f2i(p0.w));
                    if (__builtin_amdgcn_ballot_w64(d1 <= t.d3)) top3k_insert(t, d1, f2i(p1.w));
                    if (__builtin_amdgcn_ballot_w64(d2 <= t.d3)) top3k_insert(t, d2, f2i(p2.w));
                    if (__builtin_amdgcn_ballot_w64(d3 <= t.d3)) top3k_insert(t, d3, f2i(p3.w));
                }
            }
        }
    }
    if (ok) {
        const size_t o = ((size_t)bi * n + qi) * 3;
        if (dist) { dist[o] = top3k_dist(t.k1); dist[o + 1] = top3k_dist(t.k2); dist[o + 2] = top3k_dist(t.k3); }
        if (idx) { idx[o] = top3k_index(t.k1); idx[o + 1] = top3k_index(t.k2); idx[o + 2] = top3k_index(t.k3); }
    }
    if (!INTERP) return;
    float w0, w1, w2;
    fp_weights3s(top3k_dist(t.k1), top3k_dist(t.k2), top3k_dist(t.k3), w0, w1, w2);
    s_q[tid] = qi;
    s_i[tid * 3] = top3k_index(t.k1); s_i[tid * 3 + 1] = top3k_index(t.k2); s_i[tid * 3 + 2] = top3k_index(t.k3);
    s_w[tid * 3] = w0; s_w[tid * 3 + 1] = w1; s_w[tid * 3 + 2] = w2;
    __syncthreads();
    const int rows = min(NS_THREADS, n - j0);
    const int cv = c / VEC;
    const float *pb = points2 + (size_t)bi * m * c;
    for (int e = tid; e < rows * cv; e += NS_THREADS) {
        const int r = e / cv;
        const int col = (e - r * cv) * VEC;
        const vec_t a = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3] * c + col);
        const vec_t bb = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3 + 1] * c + col);
        const vec_t cc = *reinterpret_cast<const vec_t *>(pb + (size_t)s_i[r * 3 + 2] * c + col);
        const vec_t v = a * s_w[r * 3] + bb * s_w[r * 3 + 1] + cc * s_w[r * 3 + 2];   // unfused
        *reinterpret_cast<vec_t *>(out + ((size_t)bi * n + s_q[r]) * ldo + col) = v;
    }
}

// ---- workspace ---------------------------------------------------------------------------
struct NsWs {
    float *bbox;      // (b, 8)
    int *keys_q;      // (b, n)
    int *keys_k;      // (b, m)
    void *csr_q, *csr_k;
    float4 *kp;       // (b, tiles*64)
    float *tb;        // (b, tiles, 8)
    size_t bytes;
};

static size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

static NsWs ns_carve(void *ws, long long b, long long n, long long m)
{
    const long long tiles = (m + 63) / 64;
    char *p = reinterpret_cast<char *>(ws);
    size_t o = 0;
    NsWs w;
    w.bbox = reinterpret_cast<float *>(p + o); o += up256((size_t)b * 8 * 4);
    w.keys_q = reinterpret_cast<int *>(p + o); o += up256((size_t)b * n * 4);
    w.keys_k = reinterpret_cast<int *>(p + o); o += up256((size_t)b * m * 4);
    w.csr_q = p + o; o += up256(csr_ws_ints(b, NS_CELLS, n) * 4);
    w.csr_k = p + o; o += up256(csr_ws_ints(b, NS_CELLS, m) * 4);
    w.kp = reinterpret_cast<float4 *>(p + o); o += up256((size_t)b * tiles * 64 * 16);
    w.tb = reinterpret_cast<float *>(p + o); o += up256((size_t)b * tiles * 8 * 4);
    w.bytes = o;
    return w;
}

static int ns_run(int b, int n, int m, int c, const float *xyz1, const float *xyz2, const float *points2,
                  float *out, long long ldo, float *dist, int *idx, void *ws, size_t ws_bytes, hipStream_t st)
{
    NsWs w = ns_carve(ws, b, n, m);
    if (!ws || ws_bytes < w.bytes) return PRCNN_WORKSPACE;
    if (((uintptr_t)ws & 255) != 0) return PRCNN_BAD_ARG;
    const int tiles = (m + 63) / 64;
    hipLaunchKernelGGL(ns_bbox_kernel, dim3(b), dim3(1024), 0, st, n, m, xyz1, xyz2, w.bbox);
    const long long tq = (long long)b * n, tk = (long long)b * m;
    hipLaunchKernelGGL(ns_key_kernel, dim3((unsigned)((tq + 255) / 256 > 4096 ? 4096 : (tq + 255) / 256)), dim3(256), 0,
                       st, tq, n, xyz1, w.bbox, w.keys_q);
    hipLaunchKernelGGL(ns_key_kernel, dim3((unsigned)((tk + 255) / 256 > 4096 ? 4096 : (tk + 255) / 256)), dim3(256), 0,
                       st, tk, m, xyz2, w.bbox, w.keys_k);
    CsrWs cq = csr_carve(w.csr_q, b, NS_CELLS, n), ck = csr_carve(w.csr_k, b, NS_CELLS, m);
    int rc = csr_build(b, NS_CELLS, n, w.keys_q, cq, false, st);
    if (rc != PRCNN_OK) return rc;
    rc = csr_build(b, NS_CELLS, m, w.keys_k, ck, false, st);
    if (rc != PRCNN_OK) return rc;
    const long long total_tiles = (long long)b * tiles;
    hipLaunchKernelGGL(ns_tile_kernel, dim3((unsigned)((total_tiles + 3) / 4)), dim3(256), 0, st, m, tiles,
                       total_tiles, xyz2, ck.list, w.kp, w.tb);
    dim3 grid((n + NS_THREADS - 1) / NS_THREADS, b);
    if (!points2) {
        hipLaunchKernelGGL((ns_search_kernel<1, false>), grid, dim3(NS_THREADS), 0, st, n, m, 0, tiles, xyz1,
                           cq.list, w.kp, w.tb, nullptr, nullptr, 0ll, dist, idx);
    } else {
        const bool a16 = (((uintptr_t)points2 | (uintptr_t)out) & 15) == 0 && ldo % 4 == 0;
        if (c % 4 == 0 && a16)
            hipLaunchKernelGGL((ns_search_kernel<4, true>), grid, dim3(NS_THREADS), 0, st, n, m, c, tiles, xyz1,
                               cq.list, w.kp, w.tb, points2, out, ldo, dist, idx);
        else
            hipLaunchKernelGGL((ns_search_kernel<1, true>), grid, dim3(NS_THREADS), 0, st, n, m, c, tiles, xyz1,
                               cq.list, w.kp, w.tb, points2, out, ldo, dist, idx);
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
