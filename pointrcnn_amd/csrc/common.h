// Shared helpers for the gfx950 kernels of libpointrcnn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pointrcnn_hip.h"

// The last failing HIP status of the calling thread is kept for diagnostics only
// (prcnn_last_hip_error_string); it never influences a later call.
void prcnn_note_hip_error(int hip_status);
#define PRCNN_LAUNCH_CHECK()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) {                               \
            prcnn_note_hip_error((int)e__);                    \
            return PRCNN_HIP_ERROR;                            \
        }                                                      \
    } while (0)

static inline hipStream_t to_stream(prcnn_stream_t s) { return (hipStream_t)s; }

namespace prcnn {

// ---- DPP cross-lane moves (wave64; rows of 16 lanes) -----------------------------------
// dpp_ctrl encodings (gfx9): quad_perm = 0x00-0xFF, row_shr:n = 0x110+n, row_mirror = 0x140,
// row_half_mirror = 0x141, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i32(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}

// all-reduce inside each row of 16 lanes: xor1, xor2 (quad_perm), half-mirror, mirror
// `old` = INT_MIN is the identity of max, so lanes a DPP control leaves unwritten contribute
// nothing and the compiler can fold each move into a single v_max_i32_dpp.
__device__ __forceinline__ int row16_allmax_i32(int v)
{
    constexpr int ID = (int)0x80000000;
    v = max(v, dpp_i32<0xB1>(ID, v));   // quad_perm [1,0,3,2]
    v = max(v, dpp_i32<0x4E>(ID, v));   // quad_perm [2,3,0,1]
    v = max(v, dpp_i32<0x141>(ID, v));  // row_half_mirror
    v = max(v, dpp_i32<0x140>(ID, v));  // row_mirror
    return v;
}
__device__ __forceinline__ unsigned row16_allmin_u32(unsigned v)
{
    v = min(v, (unsigned)dpp_i32<0xB1>((int)v, (int)v));
    v = min(v, (unsigned)dpp_i32<0x4E>((int)v, (int)v));
    v = min(v, (unsigned)dpp_i32<0x141>((int)v, (int)v));
    v = min(v, (unsigned)dpp_i32<0x140>((int)v, (int)v));
    return v;
}
// wave-wide max of a signed int; result is wave-uniform (SGPR)
__device__ __forceinline__ int wave_max_i32(int v)
{
    constexpr int ID = (int)0x80000000;
    v = row16_allmax_i32(v);
    v = max(v, dpp_i32<0x142, 0xA>(ID, v));  // row_bcast:15 into rows 1,3
    v = max(v, dpp_i32<0x143, 0xC>(ID, v));  // row_bcast:31 into rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ int f2i(float f) { return __builtin_bit_cast(int, f); }
__device__ __forceinline__ float i2f(int i) { return __builtin_bit_cast(float, i); }

// ---- affine epilogue of the fused interpolation ------------------------------------------------
// out[q][col] = act(interp[q][col] + sum_k p1[q][k] * u[k][col] + t[col]), k < c1 <= 4: the
// "points1" half of a split FP convolution when points1 is narrow (e.g. the intensity channel),
// evaluated in the epilogue instead of being written to and re-read from HBM.  p1 == nullptr
// with c1 == 0 adds t only.
struct EpiAffine {
    const float *p1, *u, *t;
    int c1;
};

template <int VEC>
__device__ __forceinline__ void epi_affine_add(const EpiAffine &a, size_t row, int c, int col, float *v)
{
    float x[VEC];
#pragma unroll
    for (int q = 0; q < VEC; ++q) x[q] = a.t[col + q];
    for (int k = 0; k < a.c1; ++k) {
        const float pv = a.p1[row * a.c1 + k];
#pragma unroll
        for (int q = 0; q < VEC; ++q) x[q] += pv * a.u[(size_t)k * c + col + q];
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) v[q] += x[q];
}

// ---- three smallest (distance, index) pairs, branch-free -----------------------------------
// threenn_cpu (tf_interpolate.cpp:74-89) inserts with strict '<' while the index increases,
// i.e. it keeps the three smallest pairs under the lexicographic order (distance, index).
// A pair is held as one 64-bit key  (bits(distance) << 32) | index : distances are >= +0, so
// the unsigned order of the keys IS that lexicographic order (NaN sorts above +inf and never
// enters).  Insertion is three 64-bit compares and five selects with no divergence, so a
// wave pays the same whether one lane or all of them take a candidate, in any visiting order.
struct Top3K {
    unsigned long long k1, k2, k3;
    float d3;                        // distance half of k3 (screening value)
};
__device__ __forceinline__ Top3K top3k_init()
{
    const unsigned long long e = (unsigned long long)0x7f800000u << 32;      // (+inf, 0)
    Top3K t = {e, e, e, i2f(0x7f800000)};
    return t;
}
__device__ __forceinline__ void top3k_insert(Top3K &t, float d, int i)
{
    const unsigned long long c = ((unsigned long long)(unsigned)f2i(d) << 32) | (unsigned)i;
    const bool l1 = c < t.k1, l2 = c < t.k2, l3 = c < t.k3;
    t.k3 = l2 ? t.k2 : (l3 ? c : t.k3);
    t.k2 = l1 ? t.k1 : (l2 ? c : t.k2);
    t.k1 = l1 ? c : t.k1;
    t.d3 = i2f((int)(t.k3 >> 32));
}
__device__ __forceinline__ float top3k_dist(unsigned long long k) { return i2f((int)(k >> 32)); }
__device__ __forceinline__ int top3k_index(unsigned long long k) { return (int)(unsigned)k; }

}  // namespace prcnn
