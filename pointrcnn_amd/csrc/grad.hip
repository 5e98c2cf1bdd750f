// Backward ops of the PointNet++ primitives: GatherPointGrad, GroupPointGrad,
// ThreeInterpolateGrad.
//
// The reference scatters with float atomicAdd (scatteraddpointKernel
// tf_sampling_g.cu:183-192, group_point_grad_gpu tf_grouping_g.cu:61-78) or a serial CPU
// loop (threeinterpolate_grad_cpu tf_interpolate.cpp:131-153, group_point_grad_cpu
// query_ball_point.cpp:70-84).  All three are the same operation: every source entry e
// (in the CPU loop's order) adds  w[e] * src_row(e)  to destination row key[e] of a
// zero-initialised output.  Here it is an ORDERED SEGMENTED REDUCTION instead of float
// atomics:
//   1. count entries per destination row (integer atomics: order-independent),
//   2. exclusive scan per cloud,
//   3. fill each row's segment with its entry numbers (any order),
//   4. sort every segment ascending (entries are distinct integers: the result is unique),
//   5. one team of lanes per destination row adds its entries in ascending order.
// The sum order of every output element is exactly the serial CPU loop's, so the result
// is deterministic and bit-identical to the reference's CPU twins; each gradient row is
// read once, each output row written once (rows nobody points at get their zeros from
// step 5: no memset of the output), and no float atomic touches L2.
#include "common.h"
#include "csr.h"

namespace prcnn {

constexpr int GR_THREADS = 256;
constexpr int SORT_LDS_MAX = 1024;      // segment lengths up to this are ranked from LDS

typedef CsrWs ScatterWs;

size_t csr_ws_ints(long long b, long long nd, long long E)
{
    return (size_t)(b * (nd + 1) + b * nd + b * E);
}
static size_t scatter_ws_ints(long long b, long long nd, long long E) { return csr_ws_ints(b, nd, E); }

CsrWs csr_carve(void *ws, long long b, long long nd, long long E)
{
    CsrWs w;
    w.off = reinterpret_cast<int *>(ws);
    w.cursor = w.off + b * (nd + 1);
    w.list = w.cursor + b * nd;
    return w;
}
static ScatterWs carve(void *ws, long long b, long long nd, long long E) { return csr_carve(ws, b, nd, E); }

__global__ void scatter_count_kernel(long long total, long long E, int nd, const int *__restrict__ keys,
                                     int *__restrict__ off)
{
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long bi = g / E;
        const int k = keys[g];
        if (k >= 0 && k < nd) atomicAdd(&off[bi * (nd + 1) + k + 1], 1);
    }
}

// one workgroup per cloud: in-place inclusive scan of off[bi][1..nd]; cursor = row starts
__global__ __launch_bounds__(1024) void scatter_scan_kernel(int nd, int *__restrict__ off,
                                                            int *__restrict__ cursor)
{
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *o = off + (long long)blockIdx.x * (nd + 1);
    int *cur = cursor + (long long)blockIdx.x * nd;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nd; base += 1024) {
        const int k = base + tid;
        int v = k < nd ? o[k + 1] : 0;
        // wave inclusive scan
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(v, d);
            if (lane >= d) v += t;
        }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int add = carry_s;
        for (int w = 0; w < wave; ++w) add += wsum[w];
        v += add;
        if (k < nd) {
            o[k + 1] = v;                       // end of row k = start of row k+1
        }
        __syncthreads();
        if (tid == 1023) carry_s = v;
        __syncthreads();
    }
    // starts: off[k] for k in [0, nd)
    for (int k = tid; k < nd; k += 1024) cur[k] = o[k];
}

__global__ void scatter_fill_kernel(long long total, long long E, int nd, const int *__restrict__ keys,
                                    int *__restrict__ cursor, int *__restrict__ list)
{
    for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total;
         g += (long long)gridDim.x * blockDim.x) {
        const long long bi = g / E;
        const int k = keys[g];
        if (k >= 0 && k < nd) {
            const int slot = atomicAdd(&cursor[bi * nd + k], 1);
            list[bi * E + slot] = (int)(g - bi * E);
        }
    }
}

// one wave per destination row: sort its segment ascending, in place
__global__ __launch_bounds__(GR_THREADS) void scatter_sort_kernel(long long rows, long long E, int nd,
                                                                  const int *__restrict__ keys,
                                                                  const int *__restrict__ off,
                                                                  int *__restrict__ list)
{
    __shared__ int seg[GR_THREADS / 64][SORT_LDS_MAX];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long nwaves = (long long)gridDim.x * (GR_THREADS / 64);
    for (long long r = (long long)blockIdx.x * (GR_THREADS / 64) + wave; r < rows; r += nwaves) {
        const long long bi = r / nd;
        const int k = (int)(r - bi * nd);
        const int o = off[bi * (nd + 1) + k];
        const int L = off[bi * (nd + 1) + k + 1] - o;
        if (L <= 1) continue;
        int *lst = list + bi * E + o;
        if (L <= 64) {
            // rank = number of smaller entries, by broadcasting every entry once
            const int v = lane < L ? lst[lane] : 0x7fffffff;
            int rank = 0;
            for (int u = 0; u < L; ++u) rank += __shfl(v, u) < v ? 1 : 0;
            if (lane < L) lst[rank] = v;
        } else if (L <= SORT_LDS_MAX) {
            int *s = seg[wave];
            for (int t = lane; t < L; t += 64) s[t] = lst[t];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int t = lane; t < L; t += 64) {
                const int v = s[t];
                int rank = 0;
                for (int u = 0; u < L; ++u) rank += s[u] < v ? 1 : 0;
                lst[rank] = v;
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            // heavy row: rebuild the segment by scanning the cloud's keys in order
            const int *kk = keys + bi * E;
            int base = 0;
            for (long long e0 = 0; e0 < E; e0 += 64) {
                const long long e = e0 + lane;
                const bool hit = e < E && kk[e] == k;
                const unsigned long long m = __ballot(hit);
                if (hit) lst[base + __popcll(m & ((1ull << lane) - 1ull))] = (int)e;
                base += __popcll(m);
            }
        }
    }
}

// one team of T lanes per destination row; each lane owns VEC channels per pass.
// DIV: source row of entry e is e / DIV (3 for the three (idx, weight) pairs of a point).
template <int VEC, int DIV, bool WEIGHTED>
__global__ __launch_bounds__(GR_THREADS) void scatter_sum_kernel(long long rows, long long E, int nd, int c,
                                                                 int T, const int *__restrict__ off,
                                                                 const int *__restrict__ list,
                                                                 const float *__restrict__ src,
                                                                 const float *__restrict__ weight,
                                                                 float *__restrict__ dst)
{
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const int lt = threadIdx.x % T;
    const long long team0 = (blockIdx.x * (long long)GR_THREADS + threadIdx.x) / T;
    const long long nteams = (long long)gridDim.x * GR_THREADS / T;
    const long long src_rows = E / DIV;
    for (long long r = team0; r < rows; r += nteams) {
        const long long bi = r / nd;
        const int k = (int)(r - bi * nd);
        const int o = off[bi * (nd + 1) + k];
        const int L = off[bi * (nd + 1) + k + 1] - o;
        const int *lst = list + bi * E + o;
        for (int ch = lt * VEC; ch < c; ch += T * VEC) {
            vec_t acc;
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = 0.f;
            // loads four entries ahead (independent), adds strictly in order
            int t = 0;
            for (; t + 4 <= L; t += 4) {
                int e[4];
                vec_t v[4];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = lst[t + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    v[u] = *reinterpret_cast<const vec_t *>(src + (bi * src_rows + e[u] / DIV) * c + ch);
                    w[u] = WEIGHTED ? weight[bi * E + e[u]] : 1.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q)
                        acc[q] = WEIGHTED ? acc[q] + v[u][q] * w[u] : acc[q] + v[u][q];   // mul, then add: unfused
                }
            }
            for (; t < L; ++t) {
                const int e = lst[t];
                const vec_t v = *reinterpret_cast<const vec_t *>(src + (bi * src_rows + e / DIV) * c + ch);
                const float w = WEIGHTED ? weight[bi * E + e] : 1.f;
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] = WEIGHTED ? acc[q] + v[q] * w : acc[q] + v[q];
            }
            *reinterpret_cast<vec_t *>(dst + r * c + ch) = acc;
        }
    }
}

static int grid_for(long long items, int per_block, int cap)
{
    long long g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// keys (b, E) -> w.off (b, nd+1) row starts, w.list (b, E) entries grouped by row
// (ascending inside each row when sort_rows).  Keys outside [0, nd) are left out.
int csr_build(int b, int nd, long long E, const int *keys, CsrWs w, bool sort_rows, hipStream_t st)
{
    const long long rows = (long long)b * nd;
    if (hipMemsetAsync(w.off, 0, (size_t)b * (nd + 1) * sizeof(int), st) != hipSuccess) {
        prcnn_note_hip_error((int)hipGetLastError());
        return PRCNN_HIP_ERROR;
    }
    const long long total = (long long)b * E;
    if (total > 0) {
        hipLaunchKernelGGL(scatter_count_kernel, dim3(grid_for(total, GR_THREADS, 4096)), dim3(GR_THREADS), 0, st,
                           total, E, nd, keys, w.off);
    }
    hipLaunchKernelGGL(scatter_scan_kernel, dim3(b), dim3(1024), 0, st, nd, w.off, w.cursor);
    if (total > 0) {
        hipLaunchKernelGGL(scatter_fill_kernel, dim3(grid_for(total, GR_THREADS, 4096)), dim3(GR_THREADS), 0, st,
                           total, E, nd, keys, w.cursor, w.list);
        if (sort_rows)
            hipLaunchKernelGGL(scatter_sort_kernel, dim3(grid_for(rows, GR_THREADS / 64, 8192)), dim3(GR_THREADS), 0,
                               st, rows, E, nd, keys, w.off, w.list);
    }
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

// keys: (b, E) destination rows; src: (b, E/DIV, c); weight: (b, E) or null; dst: (b, nd, c)
template <int DIV, bool WEIGHTED>
static int scatter_add_ordered(int b, int nd, long long E, int c, const int *keys, const float *src,
                               const float *weight, float *dst, void *ws, size_t ws_bytes, hipStream_t st)
{
    const long long rows = (long long)b * nd;
    if (rows == 0 || c == 0) return PRCNN_OK;
    if (!dst) return PRCNN_BAD_SHAPE;
    if (E > 0x7fffffffLL) return PRCNN_UNSUPPORTED;
    if (E > 0 && (!keys || !src || (WEIGHTED && !weight))) return PRCNN_BAD_SHAPE;
    const size_t need = scatter_ws_ints(b, nd, E) * sizeof(int);
    if (!ws || ws_bytes < need) return PRCNN_WORKSPACE;
    if (((uintptr_t)ws & 3) != 0) return PRCNN_BAD_ARG;
    ScatterWs w = carve(ws, b, nd, E);
    const int rc = csr_build(b, nd, E, keys, w, true, st);
    if (rc != PRCNN_OK) return rc;
    const bool v4 = c % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0;
    const int lanes = v4 ? c / 4 : c;
    int T = 1;
    while (T < lanes && T < 64) T <<= 1;
    const int grid = grid_for(rows * T, GR_THREADS, 16384);
    if (v4)
        hipLaunchKernelGGL((scatter_sum_kernel<4, DIV, WEIGHTED>), dim3(grid), dim3(GR_THREADS), 0, st, rows, E, nd,
                           c, T, w.off, w.list, src, weight, dst);
    else
        hipLaunchKernelGGL((scatter_sum_kernel<1, DIV, WEIGHTED>), dim3(grid), dim3(GR_THREADS), 0, st, rows, E, nd,
                           c, T, w.off, w.list, src, weight, dst);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

static size_t ws_bytes_for(long long b, long long nd, long long E)
{
    if (b <= 0 || nd <= 0 || E < 0) return 0;
    return (scatter_ws_ints(b, nd, E) * sizeof(int) + 255) & ~(size_t)255;
}

}  // namespace prcnn

extern "C" size_t prcnn_gather_point_grad_workspace_bytes(int b, int n, int m)
{
    return prcnn::ws_bytes_for(b, n, m);
}

extern "C" int prcnn_gather_point_grad(int b, int n, int m, const float *out_g, const int *idx, float *inp_g,
                                       void *workspace, size_t workspace_bytes, prcnn_stream_t stream)
{
    if (b < 0 || n <= 0 || m < 0) return PRCNN_BAD_SHAPE;
    return prcnn::scatter_add_ordered<1, false>(b, n, m, 3, idx, out_g, nullptr, inp_g, workspace,
                                                workspace_bytes, to_stream(stream));
}

extern "C" size_t prcnn_group_point_grad_workspace_bytes(int b, int n, int m, int nsample)
{
    return prcnn::ws_bytes_for(b, n, (long long)m * nsample);
}

extern "C" int prcnn_group_point_grad(int b, int n, int c, int m, int nsample, const float *grad_out,
                                      const int *idx, float *grad_points, void *workspace,
                                      size_t workspace_bytes, prcnn_stream_t stream)
{
    if (b < 0 || n <= 0 || c < 0 || m < 0 || nsample < 0) return PRCNN_BAD_SHAPE;
    return prcnn::scatter_add_ordered<1, false>(b, n, (long long)m * nsample, c, idx, grad_out, nullptr,
                                                grad_points, workspace, workspace_bytes, to_stream(stream));
}

extern "C" size_t prcnn_three_interpolate_grad_workspace_bytes(int b, int n, int m)
{
    return prcnn::ws_bytes_for(b, m, 3LL * n);
}

extern "C" int prcnn_three_interpolate_grad(int b, int n, int c, int m, const float *grad_out, const int *idx,
                                            const float *weight, float *grad_points, void *workspace,
                                            size_t workspace_bytes, prcnn_stream_t stream)
{
    if (b < 0 || n < 0 || c < 0 || m <= 0) return PRCNN_BAD_SHAPE;
    return prcnn::scatter_add_ordered<3, true>(b, m, 3LL * n, c, idx, grad_out, weight, grad_points, workspace,
                                               workspace_bytes, to_stream(stream));
}
