// Fully fused set-abstraction scale for narrow layers: group -> 3 x (1x1 conv + folded BN +
// ReLU) -> max over nsample in ONE kernel; no intermediate activation ever leaves the CU.
//
// Replaces, for one radius of pointnet_sa_module[_msg] (models/pointnet_util.py:177-192 /
// :43-53,116-126), the chain  group_point, group_point, concat, conv2d x3, reduce_max.
// The generic path (mlp.hip) runs the three layers as three launches and is bound, for the
// narrow layers of the first RPN level (4 -> 16/16/32 and 4 -> 32/32/64, models/rpn.py:166-169),
// by writing and re-reading the (rows x 32) activations through HBM.  Here:
//   * all three weight matrices (+ scale/shift) sit in LDS for the whole kernel
//     (<= 32 KB: the kernel only serves layers this narrow);
//   * each WAVE owns 32 grouped rows end to end: it gathers them, runs the three layers on
//     v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain) and pools them.  A layer's output is
//     written (C/D layout -> row-major) to the wave's private LDS buffer and read back as
//     the next layer's A operand, so there is no workgroup barrier after the weight load;
//   * nsample must be 16 or 32 (a pooled group never leaves the wave's 32 rows).
// Physical k of MFMA step s / lane-half h is h*KS+s (KS = K/2): a lane's A operands of a
// whole layer are KS contiguous floats of its row (ds_read_b128 / b64), rows padded by 4
// words (conflict-free), B is read as W[h*KS+s][col] (consecutive lanes, conflict-free).
#include <stdlib.h>
#include "common.h"

namespace prcnn {

typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef float f32x2c __attribute__((ext_vector_type(2)));

// ---- duplicate-free groups ------------------------------------------------------------------
// query_ball_point pads a ball that holds cnt < nsample points with copies of its first hit
// (tf_grouping_g.cu:26-29), so the grouped tensor repeats that row nsample - cnt times and the
// max over the group does not change if the copies are dropped.  With sparse balls most of the
// grouped rows are such copies (KITTI-shaped scenes of 16384 points: 4.05 of 32 slots at radius
// 0.5, 1.56 of 16 at 0.1).  prcnn_ball_count_classes sorts the centroids of a scale into six
// classes by cnt (<= 1, 2, 4, 8, 16, 32); the chain kernels then walk, class by class, only the
// first G = 1 / 2 / 4 / 8 / 16 / 32 slots of each centroid of that class -- all its distinct rows
// plus padding up to the class size -- as tiles of 32 rows = 32 / G groups.  Same values, bit for
// bit (the dropped rows are copies of a kept one).  Without class lists the kernels see one
// class, G = nsample, holding every centroid in order: the plain layout.
constexpr int CLS_N = 6;

struct ClsArgs {
    const int *list;       // [CLS_N][stride] centroid numbers (b*m space) per class, or null
    const int *count;      // [CLS_N]
    long long stride;
};

// All of this is wave-uniform and lives in scalar registers (readfirstlane); no array is indexed
// with a run-time value (that would put it in scratch memory).
struct ClsTiles {
    int start[CLS_N + 1];         // first virtual tile of each class
    int count[CLS_N];
};

__device__ __forceinline__ void cls_setup(const ClsArgs &c, long long groups, int ns, ClsTiles &t)
{
    int acc = 0;
#pragma unroll
    for (int k = 0; k < CLS_N; ++k) {
        int cnt;
        if (c.list) cnt = __builtin_amdgcn_readfirstlane(c.count[k]);
        else cnt = (1 << k) == ns ? (int)groups : 0;
        t.count[k] = cnt;
        t.start[k] = acc;
        acc += (cnt + (32 >> k) - 1) >> (5 - k);                  // 32 >> k groups per tile
    }
    t.start[CLS_N] = acc;
}

struct ClsTile {
    int lg;                // log2 of the class's group size
    int gbase;             // first group (position in the class list) of the tile
    int count;             // groups in the class
    const int *lst;        // the class list, or null (identity)
};

__device__ __forceinline__ ClsTile cls_tile(const ClsArgs &c, const ClsTiles &t, int vt)
{
    int k = 0, st = t.start[0], cn = t.count[0];
#pragma unroll
    for (int q = 1; q < CLS_N; ++q)
        if (vt >= t.start[q]) { k = q; st = t.start[q]; cn = t.count[q]; }
    ClsTile r;
    r.lg = k;
    r.gbase = (vt - st) << (5 - k);
    r.count = cn;
    r.lst = c.list ? c.list + (long long)k * c.stride : nullptr;
    return r;
}

// centroid number (b*m space) of group position g, -1 past the end
__device__ __forceinline__ int cls_centroid(const ClsTile &ti, int g)
{
    if (g >= ti.count) return -1;
    return ti.lst ? ti.lst[g] : g;
}

// Max over the groups of a 32-row x 32-column block of the tile and store of one value per group
// and column: out[centroid*ldo + col0 + c].  The block goes through the wave's LDS tile (written
// in the MFMA C/D layout by the caller: X[row*XS + c]); lane e handles (group e / 32, column
// e % 32), so the stores of a group are 128 contiguous bytes.  (An in-register version -- G <= 4
// inside a lane, G >= 8 plus a shfl_xor -- tripled the kernels' registers: 50 -> 140 VGPRs.)
__device__ __forceinline__ void cls_pool_block(const float *X, int XS, const ClsTile &ti, int lane, int col0,
                                               int c3, float *out, long long ldo)
{
    const int G = 1 << ti.lg, ngroups = 32 >> ti.lg;
    for (int e = lane; e < 32 * ngroups; e += 64) {
        const int g = e >> 5, c = e & 31;
        const float *xr = X + (g << ti.lg) * XS + c;
        float mx = xr[0];
        for (int r = 1; r < G; ++r) mx = fmaxf(mx, xr[r * XS]);
        const int ci = cls_centroid(ti, ti.gbase + g);
        if (ci >= 0 && col0 + c < c3) out[(long long)ci * ldo + col0 + c] = mx;
    }
}

// One global atomic per class and WORKGROUP (1024 groups): returning atomics on one address
// serialise in L2 at ~50 ns each, so per-wave appends cost 50 us at 65536 groups.
__global__ __launch_bounds__(1024) void ball_classes_kernel(long long groups, const int *__restrict__ pts_cnt,
                                                            int *__restrict__ list, int *__restrict__ count)
{
    __shared__ int lcount[CLS_N], lbase[CLS_N];
    for (long long g0 = (long long)blockIdx.x * 1024; g0 < groups; g0 += (long long)gridDim.x * 1024) {
        if (threadIdx.x < CLS_N) lcount[threadIdx.x] = 0;
        __syncthreads();
        const long long g = g0 + threadIdx.x;
        int k = -1, pos = 0;
        if (g < groups) {
            const int c = pts_cnt[g];
            k = c <= 1 ? 0 : (c <= 2 ? 1 : (c <= 4 ? 2 : (c <= 8 ? 3 : (c <= 16 ? 4 : 5))));
            pos = atomicAdd(&lcount[k], 1);
        }
        __syncthreads();
        if (threadIdx.x < CLS_N) lbase[threadIdx.x] = lcount[threadIdx.x] ? atomicAdd(&count[threadIdx.x], lcount[threadIdx.x]) : 0;
        __syncthreads();
        if (k >= 0) list[k * groups + lbase[k] + pos] = (int)g;
        __syncthreads();
    }
}

struct ChainArgs {
    long long rows;            // b*m*nsample
    int n, m, ns, c, xyz_first;
    const float *xyz, *feats, *new_xyz;
    const int *idx;
    int c1, c2, c3;            // real widths
    const float *w1, *s1, *t1, *w2, *s2, *t2, *w3, *s3, *t3;
    float *out;
    long long ldo;
    ClsArgs cls;
};

constexpr int CHAIN_THREADS = 256;

// One layer for the wave's 32 rows: X[32][K] (LDS, row stride XS) x W[K][N] (LDS, stride N)
template <int K, int N, int XS>
__device__ __forceinline__ void chain_layer(const float *X, const float *W, int li, int kh,
                                            f32x16c (&acc)[N / 32])
{
    constexpr int KS = K / 2;
    constexpr int TN = N / 32;
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float af[KS];
    const float *xr = X + li * XS + kh * KS;
    if constexpr (KS % 4 == 0) {
#pragma unroll
        for (int v = 0; v < KS / 4; ++v) {
            const f32x4c t = *reinterpret_cast<const f32x4c *>(xr + 4 * v);
            af[4 * v] = t[0]; af[4 * v + 1] = t[1]; af[4 * v + 2] = t[2]; af[4 * v + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int v = 0; v < KS / 2; ++v) {
            const f32x2c t = *reinterpret_cast<const f32x2c *>(xr + 2 * v);
            af[2 * v] = t[0]; af[2 * v + 1] = t[1];
        }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float bw = W[(kh * KS + s) * N + j * 32 + li];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s], bw, acc[j], 0, 0, 0);
        }
    }
}

// scale/shift/ReLU, then C/D layout -> row-major X (C/D: col = lane&31,
// row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
template <int N, int XS>
__device__ __forceinline__ void chain_store(float *X, const float *sc, const float *sh, int li, int kh,
                                            f32x16c (&acc)[N / 32])
{
#pragma unroll
    for (int j = 0; j < N / 32; ++j) {
        const float a = sc[j * 32 + li], b = sh[j * 32 + li];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
            X[row * XS + j * 32 + li] = fmaxf(acc[j][r] * a + b, 0.f);
        }
    }
}

// C1P / C2P: widths of the first two layers padded to the 32-column MFMA tile; K1P / K2P: how
// many of those columns the NEXT layer reads as its K (16 when the real width is <= 16: the
// padded columns are zero, so skipping them halves that layer's MFMA steps).
template <int K0P, int C1P, int C2P, int C3P, int K1P = C1P, int K2P = C2P>
__global__ __launch_bounds__(CHAIN_THREADS) void sa_chain_kernel(const ChainArgs a)
{
    constexpr int CMAX = K0P > C1P ? (K0P > C2P ? K0P : C2P) : (C1P > C2P ? C1P : C2P);
    constexpr int XS = CMAX + 4;
    constexpr int WFL = K0P * C1P + K1P * C2P + K2P * C3P;      // weight floats
    constexpr int VFL = 2 * (C1P + C2P + C3P);                  // scale + shift floats
    __shared__ __attribute__((aligned(16))) float lds[WFL + VFL + 4 * 32 * XS];
    float *W1 = lds, *W2 = W1 + K0P * C1P, *W3 = W2 + K1P * C2P;
    float *S1 = W3 + K2P * C3P, *T1 = S1 + C1P, *S2 = T1 + C1P, *T2 = S2 + C2P, *S3 = T2 + C2P,
          *T3 = S3 + C3P;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    float *X = T3 + C3P + wave * 32 * XS;

    // ---- weights (zero-padded to the tile sizes) into LDS, once per workgroup ---------------
    const int k0 = a.c + 3;
    for (int e = tid; e < K0P * C1P; e += CHAIN_THREADS) {
        const int k = e / C1P, o = e - k * C1P;
        W1[e] = (k < k0 && o < a.c1) ? a.w1[k * a.c1 + o] : 0.f;
    }
    for (int e = tid; e < K1P * C2P; e += CHAIN_THREADS) {
        const int k = e / C2P, o = e - k * C2P;
        W2[e] = (k < a.c1 && o < a.c2) ? a.w2[k * a.c2 + o] : 0.f;
    }
    for (int e = tid; e < K2P * C3P; e += CHAIN_THREADS) {
        const int k = e / C3P, o = e - k * C3P;
        W3[e] = (k < a.c2 && o < a.c3) ? a.w3[k * a.c3 + o] : 0.f;
    }
    for (int e = tid; e < C1P; e += CHAIN_THREADS) { S1[e] = e < a.c1 ? a.s1[e] : 0.f; T1[e] = e < a.c1 ? a.t1[e] : 0.f; }
    for (int e = tid; e < C2P; e += CHAIN_THREADS) { S2[e] = e < a.c2 ? a.s2[e] : 0.f; T2[e] = e < a.c2 ? a.t2[e] : 0.f; }
    for (int e = tid; e < C3P; e += CHAIN_THREADS) { S3[e] = e < a.c3 ? a.s3[e] : 0.f; T3[e] = e < a.c3 ? a.t3[e] : 0.f; }
    __syncthreads();

    ClsTiles ct;
    cls_setup(a.cls, a.rows / a.ns, a.ns, ct);
    const int tiles = ct.start[CLS_N];
    for (int t = (int)blockIdx.x * 4 + wave; t < tiles; t += (int)gridDim.x * 4) {
        const ClsTile ti = cls_tile(a.cls, ct, t);
        // ---- gather: lane l < 32 builds the tile's row l = slot l % G of group l / G, as
        //      [feats, xyz-centre] (MSG) or [xyz-centre, feats] (SSG), zero-padded to K0P ----------
        if (lane < 32) {
            float v[K0P];
#pragma unroll
            for (int k = 0; k < K0P; ++k) v[k] = 0.f;
            const int ci = cls_centroid(ti, ti.gbase + (lane >> ti.lg));
            if (ci >= 0) {
                const long long bi = ci / a.m;
                const long long pi = bi * a.n + a.idx[(long long)ci * a.ns + (lane & ((1 << ti.lg) - 1))];
                const float dx = a.xyz[pi * 3 + 0] - a.new_xyz[(long long)ci * 3 + 0];
                const float dy = a.xyz[pi * 3 + 1] - a.new_xyz[(long long)ci * 3 + 1];
                const float dz = a.xyz[pi * 3 + 2] - a.new_xyz[(long long)ci * 3 + 2];
                const int xo = a.xyz_first ? 0 : a.c, fo = a.xyz_first ? 3 : 0;
#pragma unroll
                for (int k = 0; k < K0P; ++k) {
                    float val = 0.f;
                    if (k == xo) val = dx;
                    else if (k == xo + 1) val = dy;
                    else if (k == xo + 2) val = dz;
                    else if (k >= fo && k < fo + a.c) val = a.feats[pi * a.c + (k - fo)];
                    v[k] = val;
                }
            }
#pragma unroll
            for (int k = 0; k < K0P; k += 4)
                *reinterpret_cast<f32x4c *>(X + lane * XS + k) = f32x4c{v[k], v[k + 1], v[k + 2], v[k + 3]};
        }
        // wave-private buffer: LDS operations of one wave complete in order, no barrier needed
        f32x16c acc1[C1P / 32];
        chain_layer<K0P, C1P, XS>(X, W1, li, kh, acc1);
        chain_store<C1P, XS>(X, S1, T1, li, kh, acc1);
        f32x16c acc2[C2P / 32];
        chain_layer<K1P, C2P, XS>(X, W2, li, kh, acc2);
        chain_store<C2P, XS>(X, S2, T2, li, kh, acc2);
        f32x16c acc3[C3P / 32];
        chain_layer<K2P, C3P, XS>(X, W3, li, kh, acc3);
        // ---- scale/shift/ReLU, then max over each group of the tile, 32 columns at a time ----------
        if (ti.lg >= 4) {
            // groups of 16 / 32 rows: registers 0..7 (+ the other lane half) hold rows 0..15
            const int c0 = cls_centroid(ti, ti.gbase), c1 = ti.lg == 4 ? cls_centroid(ti, ti.gbase + 1) : -1;
#pragma unroll
            for (int j = 0; j < C3P / 32; ++j) {
                const int col = j * 32 + li;
                const float sc = S3[col], sh = T3[col];
                float g0 = 0.f, g1 = 0.f;                 // ReLU output is >= 0
#pragma unroll
                for (int r = 0; r < 8; ++r) g0 = fmaxf(g0, fmaxf(acc3[j][r] * sc + sh, 0.f));
#pragma unroll
                for (int r = 8; r < 16; ++r) g1 = fmaxf(g1, fmaxf(acc3[j][r] * sc + sh, 0.f));
                g0 = fmaxf(g0, __shfl_xor(g0, 32));
                g1 = fmaxf(g1, __shfl_xor(g1, 32));
                if (kh == 0 && col < a.c3) {
                    if (ti.lg == 4) {
                        if (c0 >= 0) a.out[(long long)c0 * a.ldo + col] = g0;
                        if (c1 >= 0) a.out[(long long)c1 * a.ldo + col] = g1;
                    } else if (c0 >= 0) {
                        a.out[(long long)c0 * a.ldo + col] = fmaxf(g0, g1);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < C3P / 32; ++j) {
                const float sc = S3[j * 32 + li], sh = T3[j * 32 + li];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    X[((r & 3) + 8 * (r >> 2) + 4 * kh) * XS + li] = fmaxf(acc3[j][r] * sc + sh, 0.f);
                cls_pool_block(X, XS, ti, lane, j * 32, a.c3, a.out, a.ldo);
            }
        }
    }
}

// ---- hoisted first layer + two MFMA layers + max-pool (wider scales) -----------------------
// For scales whose first convolution has been hoisted out of the grouping (a per-point table
// T = feats W_f for all scales, see prcnn_sa_group_add_layer): layer 1 is a gather of table rows
// plus the 3-column product with the relative coordinates, evaluated on the VALU straight into
// the wave's LDS tile; layers 2 and 3 run on the matrix cores as above and the max over nsample
// closes the tile.  The three launches this replaces are bound by writing and re-reading the
// (rows x 64..96) activations through HBM (K64 N96 at batch 16: 335 MB in 97 us); here nothing
// but the pooled result leaves the CU.  Serves c1 <= 64, c2 <= 64 or 96, c3 <= 128, nsample 16/32
// (the second RPN level, models/rpn.py:170-173).
struct Chain2Args {
    long long rows;            // b*m*nsample
    int n, m, ns;
    const float *xyz, *new_xyz;
    const int *idx;
    const float *table;        // (b*n, ldt): hoisted feature half of layer 1 (this scale's columns)
    long long ldt;
    int c1, c2, c3;
    const float *wx, *s1, *t1; // (3, c1) xyz rows of the first layer's weights, folded BN
    const float *w2, *s2, *t2, *w3, *s3, *t3;
    float *out;
    long long ldo;
    ClsArgs cls;
};

template <int C2P, int NW>
__global__ __launch_bounds__(NW * 64) void sa_chain2_kernel(const Chain2Args a)
{
    constexpr int C1P = 64, C3P = 128;
    constexpr int XS = (C1P > C2P ? C1P : C2P) + 4;
    constexpr int NT = NW * 64;
    __shared__ __attribute__((aligned(16))) float lds[C1P * C2P + C2P * C3P + 5 * C1P + 2 * C2P + 2 * C3P + NW * 32 * XS];
    float *W2 = lds, *W3 = W2 + C1P * C2P;
    float *WX = W3 + C2P * C3P;                 // [3][C1P]
    float *S1 = WX + 3 * C1P, *T1 = S1 + C1P, *S2 = T1 + C1P, *T2 = S2 + C2P, *S3 = T2 + C2P, *T3 = S3 + C3P;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;
    float *X = T3 + C3P + wave * 32 * XS;
    __shared__ int next_tile;
    if (tid == 0) next_tile = 0;

    for (int e = tid; e < C1P * C2P; e += NT) {
        const int k = e / C2P, o = e - k * C2P;
        W2[e] = (k < a.c1 && o < a.c2) ? a.w2[k * a.c2 + o] : 0.f;
    }
    for (int e = tid; e < C2P * C3P; e += NT) {
        const int k = e / C3P, o = e - k * C3P;
        W3[e] = (k < a.c2 && o < a.c3) ? a.w3[k * a.c3 + o] : 0.f;
    }
    for (int e = tid; e < 3 * C1P; e += NT) {
        const int k = e / C1P, o = e - k * C1P;
        WX[e] = o < a.c1 ? a.wx[k * a.c1 + o] : 0.f;
    }
    for (int e = tid; e < C1P; e += NT) { S1[e] = e < a.c1 ? a.s1[e] : 0.f; T1[e] = e < a.c1 ? a.t1[e] : 0.f; }
    for (int e = tid; e < C2P; e += NT) { S2[e] = e < a.c2 ? a.s2[e] : 0.f; T2[e] = e < a.c2 ? a.t2[e] : 0.f; }
    for (int e = tid; e < C3P; e += NT) { S3[e] = e < a.c3 ? a.s3[e] : 0.f; T3[e] = e < a.c3 ? a.t3[e] : 0.f; }
    __syncthreads();

    // this lane's 4 columns of layer 1 (fixed for the whole kernel)
    const int pc = (lane & 15) * 4;
    float wx0[4], wx1[4], wx2[4], sc1[4], sh1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        wx0[q] = WX[pc + q]; wx1[q] = WX[C1P + pc + q]; wx2[q] = WX[2 * C1P + pc + q];
        sc1[q] = S1[pc + q]; sh1[q] = T1[pc + q];
    }
    const bool tvec = (a.ldt & 3) == 0 && (((uintptr_t)a.table) & 15) == 0 && (a.c1 & 3) == 0;

    // One workgroup per CU (the weights fill most of its LDS) owns a contiguous run of tiles; its
    // waves take them one by one from an LDS counter, so a wave that has a SIMD to itself (NW not
    // a multiple of 4) simply takes more of them.
    ClsTiles ct;
    cls_setup(a.cls, a.rows / a.ns, a.ns, ct);
    const int tiles = ct.start[CLS_N];                // virtual tiles over all classes
    const int per = (tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_begin = (int)blockIdx.x * per;
    const int t_end = t_begin + per < tiles ? t_begin + per : tiles;
    // Layer-1 inputs of a tile are fetched one tile ahead into registers (the table rows come from
    // L2 / HBM: with one workgroup per CU nothing else would hide that latency).
    struct Pre {
        f32x4c tb[8];
        float rx[8], ry[8], rz[8];
        unsigned ok;             // bit it: the row of pass `it` exists
        int t;                   // (virtual) tile, -1 = none left
        ClsTile ti;
    };
    auto grab = [&]() -> int {
        int tl = 0;
        if (lane == 0) tl = atomicAdd(&next_tile, 1);
        const int t = t_begin + __builtin_amdgcn_readfirstlane(tl);
        return t < t_end ? t : -1;
    };
    auto fetch = [&](Pre &p) {
        p.t = grab();
        p.ok = 0;
        if (p.t < 0) return;
        p.ti = cls_tile(a.cls, ct, p.t);
        // lane l < 32 resolves the tile's row l = slot l % G of group l / G (point row, relative coordinates)
        long long pt = -1;
        float dx = 0.f, dy = 0.f, dz = 0.f;
        if (lane < 32) {
            const int ci = cls_centroid(p.ti, p.ti.gbase + (lane >> p.ti.lg));
            if (ci >= 0) {
                const long long bi = ci / a.m;
                pt = bi * a.n + a.idx[(long long)ci * a.ns + (lane & ((1 << p.ti.lg) - 1))];
                dx = a.xyz[pt * 3 + 0] - a.new_xyz[(long long)ci * 3 + 0];
                dy = a.xyz[pt * 3 + 1] - a.new_xyz[(long long)ci * 3 + 1];
                dz = a.xyz[pt * 3 + 2] - a.new_xyz[(long long)ci * 3 + 2];
            }
        }
        const int ptlo = (int)pt, pthi = (int)(pt >> 32);
#pragma unroll
        for (int it = 0; it < 8; ++it) {                 // 4 rows x 64 columns per pass
            const int r = it * 4 + (lane >> 4);
            const long long rp = ((long long)__shfl(pthi, r) << 32) | (unsigned)__shfl(ptlo, r);
            p.rx[it] = __shfl(dx, r); p.ry[it] = __shfl(dy, r); p.rz[it] = __shfl(dz, r);
            p.tb[it] = f32x4c{0.f, 0.f, 0.f, 0.f};
            if (rp >= 0 && pc < a.c1) {
                p.ok |= 1u << it;
                const float *src = a.table + rp * a.ldt + pc;
                if (tvec) p.tb[it] = *reinterpret_cast<const f32x4c *>(src);
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) p.tb[it][q] = pc + q < a.c1 ? src[q] : 0.f;
                }
            }
        }
    };
    Pre cur;
    fetch(cur);
    while (cur.t >= 0) {
        const ClsTile ti = cur.ti;
        // ---- layer 1 on the VALU, straight into the wave's tile ---------------------------------
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int r = it * 4 + (lane >> 4);
            f32x4c v = {0.f, 0.f, 0.f, 0.f};
            if (cur.ok & (1u << it)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float xw = __builtin_fmaf(cur.rz[it], wx2[q], __builtin_fmaf(cur.ry[it], wx1[q], cur.rx[it] * wx0[q]));
                    v[q] = fmaxf((cur.tb[it][q] + xw) * sc1[q] + sh1[q], 0.f);
                }
            }
            *reinterpret_cast<f32x4c *>(X + r * XS + pc) = v;
        }
        fetch(cur);              // next tile's loads fly while the matrix cores work on this one
        // ---- layers 2 and 3 on the matrix cores (wave-private tile, no barrier) -----------------
        f32x16c acc2[C2P / 32];
        chain_layer<C1P, C2P, XS>(X, W2, li, kh, acc2);
        chain_store<C2P, XS>(X, S2, T2, li, kh, acc2);
        f32x16c acc3[C3P / 32];
        chain_layer<C2P, C3P, XS>(X, W3, li, kh, acc3);
        if (ti.lg >= 4) {
            // groups of 16 / 32 rows: registers 0..7 (+ the other lane half) hold rows 0..15
            const int c0 = cls_centroid(ti, ti.gbase), c1 = ti.lg == 4 ? cls_centroid(ti, ti.gbase + 1) : -1;
#pragma unroll
            for (int j = 0; j < C3P / 32; ++j) {
                const int col = j * 32 + li;
                const float sc = S3[col], sh = T3[col];
                float g0 = 0.f, g1 = 0.f;                 // ReLU output is >= 0
#pragma unroll
                for (int r = 0; r < 8; ++r) g0 = fmaxf(g0, fmaxf(acc3[j][r] * sc + sh, 0.f));
#pragma unroll
                for (int r = 8; r < 16; ++r) g1 = fmaxf(g1, fmaxf(acc3[j][r] * sc + sh, 0.f));
                g0 = fmaxf(g0, __shfl_xor(g0, 32));
                g1 = fmaxf(g1, __shfl_xor(g1, 32));
                if (kh == 0 && col < a.c3) {
                    if (ti.lg == 4) {
                        if (c0 >= 0) a.out[(long long)c0 * a.ldo + col] = g0;
                        if (c1 >= 0) a.out[(long long)c1 * a.ldo + col] = g1;
                    } else if (c0 >= 0) {
                        a.out[(long long)c0 * a.ldo + col] = fmaxf(g0, g1);
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < C3P / 32; ++j) {
                const float sc = S3[j * 32 + li], sh = T3[j * 32 + li];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    X[((r & 3) + 8 * (r >> 2) + 4 * kh) * XS + li] = fmaxf(acc3[j][r] * sc + sh, 0.f);
                cls_pool_block(X, XS, ti, lane, j * 32, a.c3, a.out, a.ldo);
            }
        }
    }
}

template <int K0P, int C3P>
static int launch_chain(const ChainArgs &a, hipStream_t st)
{
    const long long tiles = (a.rows + 31) / 32;
    long long blocks = (tiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;               // persistent-style: weights are loaded once per block
    if (a.cls.list && blocks > 1024) blocks = 1024; // class lists: usually far fewer tiles than the nominal count
    if (a.c1 <= 16 && a.c2 <= 16)
        hipLaunchKernelGGL((sa_chain_kernel<K0P, 32, 32, C3P, 16, 16>), dim3((unsigned)blocks), dim3(CHAIN_THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((sa_chain_kernel<K0P, 32, 32, C3P>), dim3((unsigned)blocks), dim3(CHAIN_THREADS), 0, st, a);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

}  // namespace prcnn

static int sa_mlp3_pool_impl(int b, int n, int m, int nsample, int c, const float *xyz,
                             const float *feats, const float *new_xyz, const int *idx,
                             int xyz_first, int c1, int c2, int c3, const float *w1,
                             const float *s1, const float *t1, const float *w2,
                             const float *s2, const float *t2, const float *w3,
                             const float *s3, const float *t3, float *out, int64_t ldo,
                             const int *cls_list, const int *cls_count, prcnn_stream_t stream)
{
    using namespace prcnn;
    if ((cls_list == nullptr) != (cls_count == nullptr)) return PRCNN_BAD_SHAPE;
    if (b < 0 || n <= 0 || m < 0 || nsample <= 0 || c < 0 || c1 <= 0 || c2 <= 0 || c3 <= 0)
        return PRCNN_BAD_SHAPE;
    if (nsample != 16 && nsample != 32) return PRCNN_UNSUPPORTED;
    if (c + 3 > 8 || c1 > 32 || c2 > 32 || c3 > 128) return PRCNN_UNSUPPORTED;
    if (ldo < c3) return PRCNN_BAD_SHAPE;
    const long long rows = (long long)b * m * nsample;
    if (rows == 0) return PRCNN_OK;
    if (!xyz || !new_xyz || !idx || (c > 0 && !feats) || !out) return PRCNN_BAD_SHAPE;
    if (!w1 || !s1 || !t1 || !w2 || !s2 || !t2 || !w3 || !s3 || !t3) return PRCNN_BAD_SHAPE;
    ChainArgs a = {};
    a.rows = rows; a.n = n; a.m = m; a.ns = nsample; a.c = c; a.xyz_first = xyz_first;
    a.xyz = xyz; a.feats = feats; a.new_xyz = new_xyz; a.idx = idx;
    a.c1 = c1; a.c2 = c2; a.c3 = c3;
    a.w1 = w1; a.s1 = s1; a.t1 = t1; a.w2 = w2; a.s2 = s2; a.t2 = t2; a.w3 = w3; a.s3 = s3; a.t3 = t3;
    a.out = out; a.ldo = ldo;
    a.cls.list = cls_list; a.cls.count = cls_count; a.cls.stride = (long long)b * m;
    hipStream_t st = to_stream(stream);
    const bool k4 = c + 3 <= 4;
    if (c3 <= 32) return k4 ? launch_chain<4, 32>(a, st) : launch_chain<8, 32>(a, st);
    if (c3 <= 64) return k4 ? launch_chain<4, 64>(a, st) : launch_chain<8, 64>(a, st);
    return k4 ? launch_chain<4, 128>(a, st) : launch_chain<8, 128>(a, st);
}

extern "C" int prcnn_sa_group_mlp3_pool(int b, int n, int m, int nsample, int c, const float *xyz,
                                        const float *feats, const float *new_xyz, const int *idx,
                                        int xyz_first, int c1, int c2, int c3, const float *w1,
                                        const float *s1, const float *t1, const float *w2,
                                        const float *s2, const float *t2, const float *w3,
                                        const float *s3, const float *t3, float *out, int64_t ldo,
                                        prcnn_stream_t stream)
{
    return sa_mlp3_pool_impl(b, n, m, nsample, c, xyz, feats, new_xyz, idx, xyz_first, c1, c2, c3, w1, s1, t1, w2, s2,
                             t2, w3, s3, t3, out, ldo, nullptr, nullptr, stream);
}

extern "C" int prcnn_sa_group_mlp3_pool_classes(int b, int n, int m, int nsample, int c, const float *xyz,
                                                const float *feats, const float *new_xyz, const int *idx,
                                                int xyz_first, int c1, int c2, int c3, const float *w1,
                                                const float *s1, const float *t1, const float *w2,
                                                const float *s2, const float *t2, const float *w3,
                                                const float *s3, const float *t3, float *out, int64_t ldo,
                                                const int *cls_list, const int *cls_count,
                                                prcnn_stream_t stream)
{
    if (!cls_list || !cls_count) return PRCNN_BAD_SHAPE;
    return sa_mlp3_pool_impl(b, n, m, nsample, c, xyz, feats, new_xyz, idx, xyz_first, c1, c2, c3, w1, s1, t1, w2, s2,
                             t2, w3, s3, t3, out, ldo, cls_list, cls_count, stream);
}

extern "C" int prcnn_ball_count_classes(int64_t groups, const int *pts_cnt, int *cls_list, int *cls_count,
                                        prcnn_stream_t stream)
{
    if (groups < 0) return PRCNN_BAD_SHAPE;
    if (!cls_count) return PRCNN_BAD_SHAPE;
    hipStream_t st = to_stream(stream);
    if (hipMemsetAsync(cls_count, 0, 8 * sizeof(int), st) != hipSuccess) return PRCNN_HIP_ERROR;
    if (groups == 0) return PRCNN_OK;
    if (!pts_cnt || !cls_list) return PRCNN_BAD_SHAPE;
    long long blocks = (groups + 1023) / 1024;
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(prcnn::ball_classes_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, (long long)groups, pts_cnt,
                       cls_list, cls_count);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

static int sa_add_mlp2_pool_impl(int b, int n, int m, int nsample, const float *xyz,
                                 const float *new_xyz, const int *idx, const float *table,
                                 int64_t ldt, int c1, int c2, int c3, const float *w_xyz,
                                 const float *s1, const float *t1, const float *w2,
                                 const float *s2, const float *t2, const float *w3,
                                 const float *s3, const float *t3, float *out, int64_t ldo,
                                 const int *cls_list, const int *cls_count, prcnn_stream_t stream)
{
    using namespace prcnn;
    if ((cls_list == nullptr) != (cls_count == nullptr)) return PRCNN_BAD_SHAPE;
    if (b < 0 || n <= 0 || m < 0 || nsample <= 0 || c1 <= 0 || c2 <= 0 || c3 <= 0) return PRCNN_BAD_SHAPE;
    if (nsample != 16 && nsample != 32) return PRCNN_UNSUPPORTED;
    if (c1 > 64 || c2 > 96 || c3 > 128) return PRCNN_UNSUPPORTED;
    if (ldo < c3 || ldt < c1) return PRCNN_BAD_SHAPE;
    const long long rows = (long long)b * m * nsample;
    if (rows == 0) return PRCNN_OK;
    if (!xyz || !new_xyz || !idx || !table || !out) return PRCNN_BAD_SHAPE;
    if (!w_xyz || !s1 || !t1 || !w2 || !s2 || !t2 || !w3 || !s3 || !t3) return PRCNN_BAD_SHAPE;
    Chain2Args a = {};
    a.rows = rows; a.n = n; a.m = m; a.ns = nsample; a.xyz = xyz; a.new_xyz = new_xyz; a.idx = idx;
    a.table = table; a.ldt = ldt; a.c1 = c1; a.c2 = c2; a.c3 = c3;
    a.wx = w_xyz; a.s1 = s1; a.t1 = t1; a.w2 = w2; a.s2 = s2; a.t2 = t2; a.w3 = w3; a.s3 = s3; a.t3 = t3;
    a.out = out; a.ldo = ldo;
    a.cls.list = cls_list; a.cls.count = cls_count; a.cls.stride = (long long)b * m;
    hipStream_t st = to_stream(stream);
    const long long tiles = (rows + 31) / 32;
    static const int nw_env = getenv("PRCNN_CHAIN2_NW") ? atoi(getenv("PRCNN_CHAIN2_NW")) : 0;   // tuning aid
    // At most 128 workgroups, i.e. half the CUs: a workgroup takes ~150 KB of a CU's LDS, so while one is resident
    // no other kernel's workgroup fits beside it, although its 6 - 12 waves leave most of the CU's issue slots idle.
    // With another batch in flight on a second stream (the bench's schedule) 256 workgroups stall that stream for
    // the launch's whole duration: 10 986 / 11 056 scenes/s with 256, 11 212 - 11 272 with 128 (64 .. 160 alike;
    // the same cap on sa_chain, whose workgroups are small, gains nothing).  Alone the launch is slower with 128
    // (tools/step_table.py); PRCNN_CHAIN2_BLOCKS overrides.
    // Only with class lists (few tiles per workgroup either way); the plain layout has eight times the tiles and
    // keeps one workgroup per CU (no-dedup step: 6217 scenes/s with 256, 5998 with 128).
    static const int blocks_env = getenv("PRCNN_CHAIN2_BLOCKS") ? atoi(getenv("PRCNN_CHAIN2_BLOCKS")) : 0;
    const int max_blocks = blocks_env > 0 ? blocks_env : (cls_list ? 128 : 256);
#define PRCNN_CHAIN2_LAUNCH(C2P_, NW_)                                                                  \
    do {                                                                                                \
        long long blocks = (tiles + NW_ - 1) / NW_;                                                     \
        if (blocks > max_blocks) blocks = max_blocks; /* weights are loaded once per workgroup */      \
        hipLaunchKernelGGL((sa_chain2_kernel<C2P_, NW_>), dim3((unsigned)blocks), dim3(NW_ * 64), 0, st, a); \
    } while (0)
    if (c2 <= 64) {
        // 48 KB of weights + NW x 8.7 KB tiles: one workgroup per CU
        if (nw_env == 4) PRCNN_CHAIN2_LAUNCH(64, 4);
        else if (nw_env == 12) PRCNN_CHAIN2_LAUNCH(64, 12);
        else if (nw_env == 8) PRCNN_CHAIN2_LAUNCH(64, 8);
        else PRCNN_CHAIN2_LAUNCH(64, 12);   // measured (b16, 262144 rows): 4 waves 112 us, 8: 89, 12: 81
    } else {
        // 72 KB of weights + NW x 12.8 KB tiles
        if (nw_env == 4) PRCNN_CHAIN2_LAUNCH(96, 4);
        else PRCNN_CHAIN2_LAUNCH(96, 6);
    }
#undef PRCNN_CHAIN2_LAUNCH
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_sa_group_add_mlp2_pool(int b, int n, int m, int nsample, const float *xyz,
                                            const float *new_xyz, const int *idx, const float *table,
                                            int64_t ldt, int c1, int c2, int c3, const float *w_xyz,
                                            const float *s1, const float *t1, const float *w2,
                                            const float *s2, const float *t2, const float *w3,
                                            const float *s3, const float *t3, float *out, int64_t ldo,
                                            prcnn_stream_t stream)
{
    return sa_add_mlp2_pool_impl(b, n, m, nsample, xyz, new_xyz, idx, table, ldt, c1, c2, c3, w_xyz, s1, t1, w2, s2, t2,
                                 w3, s3, t3, out, ldo, nullptr, nullptr, stream);
}

extern "C" int prcnn_sa_group_add_mlp2_pool_classes(int b, int n, int m, int nsample, const float *xyz,
                                                    const float *new_xyz, const int *idx, const float *table,
                                                    int64_t ldt, int c1, int c2, int c3, const float *w_xyz,
                                                    const float *s1, const float *t1, const float *w2,
                                                    const float *s2, const float *t2, const float *w3,
                                                    const float *s3, const float *t3, float *out, int64_t ldo,
                                                    const int *cls_list, const int *cls_count,
                                                    prcnn_stream_t stream)
{
    if (!cls_list || !cls_count) return PRCNN_BAD_SHAPE;
    return sa_add_mlp2_pool_impl(b, n, m, nsample, xyz, new_xyz, idx, table, ldt, c1, c2, c3, w_xyz, s1, t1, w2, s2, t2,
                                 w3, s3, t3, out, ldo, cls_list, cls_count, stream);
}
