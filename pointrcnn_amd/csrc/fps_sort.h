// Grid sort of one cloud inside a workgroup, shared by the samplers that cut a cloud into buckets of
// neighbouring points (fps_pruned_kernel in fps.hip, fb_sort_kernel in fps_bucket.hip).  Counting
// sort by a 14-bit anisotropic Morton key over the cloud's bounding box, histogram in LDS; the
// order inside a cell is arbitrary and nothing may depend on it.
#pragma once
#include "common.h"

namespace prcnn {

constexpr int FPS_SORT_KEY_BITS = 14;
constexpr int FPS_SORT_CELLS = 1 << FPS_SORT_KEY_BITS;

// LDS the sort needs, declared by the calling kernel (NW = waves of the workgroup)
template <int NW>
struct FpsSortLds {
    int hist[FPS_SORT_CELLS];
    float red[NW][6];
    float sbox[4];
    int sbits[3];
    int wsum[NW];
};

// order[r] = index (in `data`, < n <= 65536) of the point of sorted rank r.  All NW * 64 threads call.
template <int NW>
__device__ __forceinline__ void fps_grid_sort(const float *__restrict__ data, int n, FpsSortLds<NW> &s,
                                              unsigned short *order)
{
    constexpr int NT = NW * 64;
    static_assert(FPS_SORT_CELLS % NT == 0, "cells are dealt evenly to the threads");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = tid; i < n; i += NT)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = data[i * 3 + d];
            lo[d] = fminf(lo[d], v);
            hi[d] = fmaxf(hi[d], v);
        }
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
        }
    if (lane == 0)
        for (int d = 0; d < 3; ++d) { s.red[wave][d] = lo[d]; s.red[wave][3 + d] = hi[d]; }
    for (int c = tid; c < FPS_SORT_CELLS; c += NT) s.hist[c] = 0;
    __syncthreads();
    if (tid == 0) {
        float ext[3], mx = 0.f;
        for (int d = 0; d < 3; ++d) {
            float l = s.red[0][d], h = s.red[0][3 + d];
            for (int w = 1; w < NW; ++w) { l = fminf(l, s.red[w][d]); h = fmaxf(h, s.red[w][3 + d]); }
            ext[d] = h - l;
            if (!(ext[d] >= 0.f) || !(ext[d] < INFINITY)) ext[d] = 0.f;
            mx = fmaxf(mx, ext[d]);
            s.sbox[d] = l;
        }
        int bits[3] = {0, 0, 0};
        float inv = 0.f;
        // (an extent so small that mx / 1024 underflows to zero keeps every point in cell 0: the loop
        // below would never end on cell == 0)
        if (mx > 0.f && mx / 1024.f > 0.f) {
            float cell = mx / 1024.f;
            for (;;) {
                int tot = 0;
                for (int d = 0; d < 3; ++d) {
                    const float cells = ext[d] / cell;
                    int bt = 0;
                    while (bt < 10 && (float)(1 << bt) < cells) ++bt;
                    bits[d] = bt;
                    tot += bt;
                }
                if (tot <= FPS_SORT_KEY_BITS) break;
                cell *= 2.f;
            }
            inv = 1.f / cell;
        }
        s.sbox[3] = inv;
        for (int d = 0; d < 3; ++d) s.sbits[d] = bits[d];
    }
    __syncthreads();
    const float b0 = s.sbox[0], b1 = s.sbox[1], b2 = s.sbox[2], inv = s.sbox[3];
    const int t0 = s.sbits[0], t1 = s.sbits[1], t2 = s.sbits[2];
    auto key_of = [&](int i) {
        const float *p = data + i * 3;
        const float f0 = (p[0] - b0) * inv, f1 = (p[1] - b1) * inv, f2 = (p[2] - b2) * inv;
        const int q0 = f0 > 0.f ? (int)fminf(f0, (float)((1 << t0) - 1)) : 0;
        const int q1 = f1 > 0.f ? (int)fminf(f1, (float)((1 << t1) - 1)) : 0;
        const int q2 = f2 > 0.f ? (int)fminf(f2, (float)((1 << t2) - 1)) : 0;
        int key = 0, pos = 0;
        for (int lv = 0; lv < 10; ++lv) {
            if (lv < t0) { key |= ((q0 >> lv) & 1) << pos; ++pos; }
            if (lv < t1) { key |= ((q1 >> lv) & 1) << pos; ++pos; }
            if (lv < t2) { key |= ((q2 >> lv) & 1) << pos; ++pos; }
        }
        return key & (FPS_SORT_CELLS - 1);          // (always in range; the mask makes that evident)
    };
    for (int i = tid; i < n; i += NT) atomicAdd(&s.hist[key_of(i)], 1);
    __syncthreads();
    constexpr int PER = FPS_SORT_CELLS / NT;
    int sum = 0;
    for (int c = 0; c < PER; ++c) sum += s.hist[tid * PER + c];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) s.wsum[wave] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int w = 0; w < wave; ++w) run += s.wsum[w];
    for (int c = 0; c < PER; ++c) {
        const int h = s.hist[tid * PER + c];
        s.hist[tid * PER + c] = run;
        run += h;
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) order[atomicAdd(&s.hist[key_of(i)], 1)] = (unsigned short)i;
    __syncthreads();
}

}  // namespace prcnn
