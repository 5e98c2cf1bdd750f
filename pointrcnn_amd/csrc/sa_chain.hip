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
#include "common.h"

namespace prcnn {

typedef float f32x16c __attribute__((ext_vector_type(16)));
typedef float f32x4c __attribute__((ext_vector_type(4)));
typedef float f32x2c __attribute__((ext_vector_type(2)));

struct ChainArgs {
    long long rows;            // b*m*nsample
    int n, m, ns, c, xyz_first;
    const float *xyz, *feats, *new_xyz;
    const int *idx;
    int c1, c2, c3;            // real widths
    const float *w1, *s1, *t1, *w2, *s2, *t2, *w3, *s3, *t3;
    float *out;
    long long ldo;
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

    const long long tiles = (a.rows + 31) / 32;
    for (long long t = (long long)blockIdx.x * 4 + wave; t < tiles; t += (long long)gridDim.x * 4) {
        const long long row0 = t * 32;
        // ---- gather: lane l < 32 builds grouped row row0+l = [feats, xyz-centre] (MSG) or
        //      [xyz-centre, feats] (SSG), zero-padded to K0P, in the wave's X buffer ------------
        if (lane < 32) {
            const long long r = row0 + lane;
            float v[K0P];
#pragma unroll
            for (int k = 0; k < K0P; ++k) v[k] = 0.f;
            if (r < a.rows) {
                const long long bi = r / ((long long)a.m * a.ns);
                const long long pi = bi * a.n + a.idx[r];
                const long long ci = r / a.ns;
                const float dx = a.xyz[pi * 3 + 0] - a.new_xyz[ci * 3 + 0];
                const float dy = a.xyz[pi * 3 + 1] - a.new_xyz[ci * 3 + 1];
                const float dz = a.xyz[pi * 3 + 2] - a.new_xyz[ci * 3 + 2];
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
        // ---- scale/shift/ReLU + max over nsample (16: two groups per tile, 32: one) -----------
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
                if (a.ns == 16) {
                    const long long o0 = row0 / 16;
                    if (row0 < a.rows) a.out[o0 * a.ldo + col] = g0;
                    if (row0 + 16 < a.rows) a.out[(o0 + 1) * a.ldo + col] = g1;
                } else {
                    if (row0 < a.rows) a.out[(row0 / 32) * a.ldo + col] = fmaxf(g0, g1);
                }
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
    if (a.c1 <= 16 && a.c2 <= 16)
        hipLaunchKernelGGL((sa_chain_kernel<K0P, 32, 32, C3P, 16, 16>), dim3((unsigned)blocks), dim3(CHAIN_THREADS), 0, st, a);
    else
        hipLaunchKernelGGL((sa_chain_kernel<K0P, 32, 32, C3P>), dim3((unsigned)blocks), dim3(CHAIN_THREADS), 0, st, a);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

}  // namespace prcnn

extern "C" int prcnn_sa_group_mlp3_pool(int b, int n, int m, int nsample, int c, const float *xyz,
                                        const float *feats, const float *new_xyz, const int *idx,
                                        int xyz_first, int c1, int c2, int c3, const float *w1,
                                        const float *s1, const float *t1, const float *w2,
                                        const float *s2, const float *t2, const float *w3,
                                        const float *s3, const float *t3, float *out, int64_t ldo,
                                        prcnn_stream_t stream)
{
    using namespace prcnn;
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
    hipStream_t st = to_stream(stream);
    const bool k4 = c + 3 <= 4;
    if (c3 <= 32) return k4 ? launch_chain<4, 32>(a, st) : launch_chain<8, 32>(a, st);
    if (c3 <= 64) return k4 ? launch_chain<4, 64>(a, st) : launch_chain<8, 64>(a, st);
    return k4 ? launch_chain<4, 128>(a, st) : launch_chain<8, 128>(a, st);
}
