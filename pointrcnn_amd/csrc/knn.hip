// kNN grouping: SelectionSort / knn_point.
//
// Reference: knn_point (models/tf_ops/grouping/tf_grouping.py:49-74) materialises the
// (b,m,n) squared-distance matrix with TF ops and runs selection_sort_gpu
// (tf_grouping_g.cu:83-123; CPU twin selection_sort_cpu, selection_sort.cpp:20-63): k steps of
// "find the FIRST position of the minimum of positions s..n-1, swap it to position s".
// The result is the k smallest values in ascending order, but the order of EQUAL values is
// whatever the swaps produce (an element displaced from the front lands where a minimum was),
// so it is reproduced here exactly, without the O(k*n) serial scan and without the matrix:
//   1. radix select (4 passes of 8 bits over order-preserving keys) finds the k-th smallest
//      value v and how many of the picks have value v (r);
//   2. one ordered pass collects the only elements the k steps can ever touch: positions
//      0..k-1, every value < v, and the first r values == v at positions >= k (at most 3k
//      elements), keeping their relative order;
//   3. the k selection steps run on that short list (wave-parallel first-position argmin).
// Step 3 on the compacted list is step-for-step the reference's loop: elements outside the
// list are never a minimum of a remaining range and never displaced.
// One wave per query row; distances (dx*dx+dy*dy)+dz*dz with d = dataset - query, unfused.
#include "common.h"

namespace prcnn {

constexpr int KNN_THREADS = 256;
constexpr int KNN_WAVES = KNN_THREADS / 64;
constexpr int KNN_KMAX = 128;
constexpr int KNN_CAP = 3 * KNN_KMAX;
constexpr int KNN_TILE = 2048;

// order-preserving key of a float for unsigned compares; -0 and +0 share a key
__device__ __forceinline__ unsigned knn_key(float f)
{
    unsigned u = (unsigned)f2i(f);
    if (u == 0x80000000u) u = 0;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ int wave_min_i32(int v) { return ~wave_max_i32(~v); }
__device__ __forceinline__ unsigned wave_min_u32(unsigned v)
{
    return (unsigned)wave_min_i32((int)(v ^ 0x80000000u)) ^ 0x80000000u;
}

// FROM_XYZ: rows are (query j of cloud bi) against xyz1; else rows of a dist matrix (b*m, n).
// FULL: also write the complete permuted (n-long) rows like the reference op.
template <bool FROM_XYZ, bool FULL>
__global__ __launch_bounds__(KNN_THREADS) void knn_select_kernel(int n, int m, int k,
                                                                 const float *__restrict__ xyz1,
                                                                 const float *__restrict__ xyz2,
                                                                 const float *__restrict__ dist,
                                                                 float *__restrict__ out_val,
                                                                 int *__restrict__ out_idx)
{
    __shared__ float tile[FROM_XYZ ? KNN_TILE * 3 : 1];
    __shared__ int hist[KNN_WAVES][256];
    __shared__ float bval[KNN_WAVES][KNN_CAP];
    __shared__ int bidx[KNN_WAVES][KNN_CAP];      // original index of the element now in the slot
    __shared__ int bpos[KNN_WAVES][KNN_CAP];      // position of the slot in the full row
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bi = blockIdx.y;
    const int j = blockIdx.x * KNN_WAVES + wave;              // query row inside the cloud
    const bool rowok = j < m;
    const size_t row = (size_t)bi * m + (rowok ? j : 0);
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (FROM_XYZ) {
        const float *q = xyz2 + row * 3;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
    const float *pts = FROM_XYZ ? xyz1 + (size_t)bi * n * 3 : nullptr;
    const float *drow = FROM_XYZ ? nullptr : dist + row * n;
    const unsigned long long lt = (1ull << lane) - 1ull;

    // value of position t0 + lane of the current tile / row
    auto value = [&](int base, int t) -> float {
        if (FROM_XYZ) {
            const float *p = tile + (t - base) * 3;
            const float dx = p[0] - qx, dy = p[1] - qy, dz = p[2] - qz;
            return dx * dx + dy * dy + dz * dz;
        }
        return drow[t];
    };
    auto stage = [&](int base, int tn) {
        if (FROM_XYZ) {
            __syncthreads();
            for (int e = tid; e < tn * 3; e += KNN_THREADS) tile[e] = pts[(size_t)base * 3 + e];
            __syncthreads();
        }
    };
    const int step = FROM_XYZ ? KNN_TILE : n;

    // ---- 1. radix select of the k-th smallest key ------------------------------------------
    unsigned prefix = 0;
    int kk = k - 1;                                             // 0-based rank still to resolve
    for (int pass = 3; pass >= 0; --pass) {
        for (int e = lane; e < 256; e += 64) hist[wave][e] = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < n; base += step) {
            const int tn = min(step, n - base);
            stage(base, tn);
            for (int t0 = base; t0 < base + tn; t0 += 64) {
                const int t = t0 + lane;
                if (t < base + tn) {
                    const unsigned u = knn_key(value(base, t));
                    const bool match = pass == 3 || (u >> (8 * (pass + 1))) == prefix;
                    if (match) atomicAdd(&hist[wave][(u >> (8 * pass)) & 255u], 1);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // lane l owns bins 4l..4l+3
        const int h0 = hist[wave][lane * 4], h1 = hist[wave][lane * 4 + 1], h2 = hist[wave][lane * 4 + 2],
                  h3 = hist[wave][lane * 4 + 3];
        const int s = h0 + h1 + h2 + h3;
        int incl = s;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int v = __shfl_up(incl, d);
            if (lane >= d) incl += v;
        }
        const int ex = incl - s;
        const bool mine = ex <= kk && kk < incl;                // exactly one lane
        int bin = 0, below = 0;
        if (mine) {
            int c = ex;
            if (kk < c + h0) { bin = lane * 4; below = c; }
            else if (kk < c + h0 + h1) { bin = lane * 4 + 1; below = c + h0; }
            else if (kk < c + h0 + h1 + h2) { bin = lane * 4 + 2; below = c + h0 + h1; }
            else { bin = lane * 4 + 3; below = c + h0 + h1 + h2; }
        }
        const unsigned long long who = __ballot(mine);
        const int src = __builtin_ctzll(who | (1ull << 63));
        bin = __shfl(bin, src);
        below = __shfl(below, src);
        kk -= below;
        prefix = (prefix << 8) | (unsigned)bin;
        __builtin_amdgcn_wave_barrier();
    }
    const unsigned vk = prefix;                                 // key of the k-th smallest value
    const int r = kk + 1;                                       // picks that have exactly this value

    // ---- 2. ordered collection of the elements the k steps can touch -------------------------
    int L = 0, ties = 0;
    for (int base = 0; base < n; base += step) {
        const int tn = min(step, n - base);
        stage(base, tn);
        for (int t0 = base; t0 < base + tn; t0 += 64) {
            const int t = t0 + lane;
            const bool in = t < base + tn;
            const float v = in ? value(base, t) : 0.f;
            const unsigned u = knn_key(v);
            const bool tie = in && t >= k && u == vk;
            const unsigned long long tm = __ballot(tie);
            const bool take = in && (t < k || u < vk || (tie && ties + __popcll(tm & lt) < r));
            const unsigned long long km = __ballot(take);
            if (take) {
                const int slot = L + __popcll(km & lt);
                bval[wave][slot] = v;
                bidx[wave][slot] = t;
                bpos[wave][slot] = t;
            }
            L += __popcll(km);
            ties += __popcll(tm);
            if (FULL && in && rowok) {                          // untouched positions: identity
                out_val[row * n + t] = v;
                out_idx[row * n + t] = t;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    __builtin_amdgcn_wave_barrier();

    // ---- 3. the k selection steps on the short list -------------------------------------------
    for (int s = 0; s < k; ++s) {
        unsigned bk = 0xffffffffu;
        int bj = 0x7fffffff;
        for (int jj = s + lane; jj < L; jj += 64) {
            const unsigned u = knn_key(bval[wave][jj]);
            if (u < bk) { bk = u; bj = jj; }                    // strict '<': earliest position per lane
        }
        const unsigned mk = wave_min_u32(bk);
        const int jmin = wave_min_i32(bk == mk ? bj : 0x7fffffff);
        if (jmin != s && lane == 0) {                           // selection_sort.cpp:49-56
            const float tv = bval[wave][jmin]; bval[wave][jmin] = bval[wave][s]; bval[wave][s] = tv;
            const int ti = bidx[wave][jmin]; bidx[wave][jmin] = bidx[wave][s]; bidx[wave][s] = ti;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (!rowok) return;
    if (FULL) {
        for (int jj = lane; jj < L; jj += 64) {
            out_val[row * n + bpos[wave][jj]] = bval[wave][jj];
            out_idx[row * n + bpos[wave][jj]] = bidx[wave][jj];
        }
    } else {
        for (int s = lane; s < k; s += 64) {
            out_val[row * k + s] = bval[wave][s];
            out_idx[row * k + s] = bidx[wave][s];
        }
    }
}

static int knn_check(int b, int n, int m, int k)
{
    if (k <= 0) return PRCNN_BAD_ARG;                             // tf_grouping.cpp:113
    if (b < 0 || n <= 0 || m < 0) return PRCNN_BAD_SHAPE;
    if (k > n) return PRCNN_BAD_ARG;                              // the reference loop would run off the row
    if (k > KNN_KMAX || b > 65535) return PRCNN_UNSUPPORTED;
    return PRCNN_OK;
}

}  // namespace prcnn

extern "C" int prcnn_selection_sort(int b, int n, int m, int k, const float *dist, int *outi, float *out,
                                    prcnn_stream_t stream)
{
    using namespace prcnn;
    const int rc = knn_check(b, n, m, k);
    if (rc != PRCNN_OK) return rc;
    if (b == 0 || m == 0) return PRCNN_OK;
    if (!dist || !outi || !out) return PRCNN_BAD_SHAPE;
    dim3 grid((m + KNN_WAVES - 1) / KNN_WAVES, b);
    hipLaunchKernelGGL((knn_select_kernel<false, true>), grid, dim3(KNN_THREADS), 0, to_stream(stream), n, m, k,
                       nullptr, nullptr, dist, out, outi);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_knn_point(int b, int n, int m, int k, const float *xyz1, const float *xyz2, float *val,
                               int *idx, prcnn_stream_t stream)
{
    using namespace prcnn;
    const int rc = knn_check(b, n, m, k);
    if (rc != PRCNN_OK) return rc;
    if (b == 0 || m == 0) return PRCNN_OK;
    if (!xyz1 || !xyz2 || !val || !idx) return PRCNN_BAD_SHAPE;
    dim3 grid((m + KNN_WAVES - 1) / KNN_WAVES, b);
    hipLaunchKernelGGL((knn_select_kernel<true, false>), grid, dim3(KNN_THREADS), 0, to_stream(stream), n, m, k,
                       xyz1, xyz2, nullptr, val, idx);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}
