// Shared MLP (1x1 conv + folded BN + ReLU) with optional fused grouping prologue and
// fused max-pool epilogue, on the gfx950 matrix cores.
//
// Replaces the conv2d/batch_norm/relu/reduce_max chain of the reference's
// pointnet_sa_module[_msg] / pointnet_fp_module (models/pointnet_util.py:116-126,187-192,
// 222-226; models/tf_util.py:120-185,512-531), which runs cuDNN 1x1 convolutions over a
// materialised (b,m,nsample,C) tensor.  Here:
//   * precision 0: v_mfma_f32_32x32x2_f32 -- exact fp32 FMA chain (1e-5 tier);
//     precision 1: v_mfma_f32_32x32x16_bf16 with fp32 accumulation (1e-2 tier).
//   * the A operand (rows x K) is either a dense row-major matrix or is formed on the
//     fly from (xyz, feats, new_xyz, idx): the grouped tensor of group_point never
//     touches HBM (GATHER prologue);
//   * the epilogue applies scale/shift/ReLU and, if pool > 1, max-reduces consecutive
//     groups of `pool` rows before anything is stored (POOL epilogue).
// Workgroup = 256 threads = 4 waves; block tile 128 x BN (BN = 32, 64 or 128), K in
// chunks of 16 (fp32) / 32 (bf16) staged through LDS with register prefetch of the next
// chunk.  Inside a chunk the physical k of MFMA step s / lane-half h is h*8+s, so a
// lane's A operands of a whole chunk are 8 contiguous floats (2 x ds_read_b128);
// A rows are padded to 20 floats (conflict-free b128 reads), B is [k][BN] (conflict-free
// b32 reads).
#include <stdlib.h>
#include "common.h"

namespace prcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct MlpArgs {
    long long rows;
    int cin, cout;
    const float *x;
    long long ldx;
    const float *w, *scale, *shift;
    int relu, pool;
    float *out;
    long long ldo;
    // gather prologue
    int n, m, ns, c, xyz_first;
    const float *xyz, *feats, *new_xyz;
    const int *idx;
    // gather prologue only: per-point rows added to the accumulators before scale/shift
    // (the feature half of a first SA layer, evaluated once per point instead of once per
    // grouped row): acc[row][col] += add[(bi*n + idx[row]) * ldadd + col]
    const float *add;
    long long ldadd;
    // compacted rows (duplicate-free groups, see prcnn_ball_compact_rows): the number of rows is read
    // from device memory (rows is then the capacity the grid was sized for), and the gather prologue
    // takes a row's point row / centroid from explicit arrays instead of the (b,m,nsample) layout
    const int *rows_dev;
    const int *rowpi, *rowci;
};

constexpr int MLP_THREADS = 256;
constexpr int DYN_ROW_BLOCKS = 96;     // grid rows of the data-dependent launches (an empty workgroup costs ~0.1 us)
constexpr int BM = 128;

// ---- A-operand element sources ---------------------------------------------------------
struct RowSrc {
    bool valid;
    long long pi;    // dense: row index; gather: global point index bi*n + idx
    long long ci;    // gather: centroid index (row / ns)
};

template <bool GATHER>
__device__ __forceinline__ RowSrc make_row(const MlpArgs &a, long long r, long long rows)
{
    RowSrc s;
    s.valid = r < rows;
    s.pi = 0; s.ci = 0;
    if (s.valid) {
        if (GATHER && a.rowpi) {
            s.pi = a.rowpi[r];
            s.ci = a.rowci[r];
        } else if (GATHER) {
            const long long bi = r / ((long long)a.m * a.ns);
            s.pi = bi * a.n + a.idx[r];
            s.ci = r / a.ns;
        } else {
            s.pi = r;
        }
    }
    return s;
}

template <bool GATHER>
__device__ __forceinline__ float load_a1(const MlpArgs &a, const RowSrc &s, int col)
{
    if (!s.valid || col >= a.cin) return 0.f;
    if (!GATHER) return a.x[s.pi * a.ldx + col];
    int fc, xc;           // feature column / xyz column (one of them valid)
    if (a.xyz_first) { xc = col; fc = col - 3; } else { fc = col; xc = col - a.c; }
    if (fc >= 0 && fc < a.c) return a.feats[s.pi * a.c + fc];
    return a.xyz[s.pi * 3 + xc] - a.new_xyz[s.ci * 3 + xc];
}

// 4 consecutive columns starting at col (col % 4 == 0)
template <bool GATHER>
__device__ __forceinline__ f32x4 load_a4(const MlpArgs &a, const RowSrc &s, int col, bool vec_ok)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!s.valid || col >= a.cin) return v;
    if (!GATHER) {
        if (vec_ok && col + 4 <= a.cin) return *reinterpret_cast<const f32x4 *>(a.x + s.pi * a.ldx + col);
    } else {
        if (vec_ok && !a.xyz_first && col + 4 <= a.c)
            return *reinterpret_cast<const f32x4 *>(a.feats + s.pi * a.c + col);
    }
    v[0] = load_a1<GATHER>(a, s, col);
    v[1] = load_a1<GATHER>(a, s, col + 1);
    v[2] = load_a1<GATHER>(a, s, col + 2);
    v[3] = load_a1<GATHER>(a, s, col + 3);
    return v;
}

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi)
{
    // round-to-nearest-even via the hardware conversion (keeps NaN a NaN)
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 f = {lo, hi};
    bf2 b = __builtin_convertvector(f, bf2);
    return __builtin_bit_cast(unsigned, b);
}

// ---- kernel ---------------------------------------------------------------------------
// WM x WN waves (WM*WN == 4), each wave owns TM x TN MFMA tiles of 32x32.
template <int WM, int WN, int TM, int TN, bool GATHER, bool BF16, int KB, bool ADD = false>
__device__ __forceinline__ void mlp_tile(const MlpArgs &a, const long long row_block, const long long rows)
{
    constexpr int BN = WN * TN * 32;
    constexpr int BK = BF16 ? 32 : KB;             // fp32: 16 (narrow inputs) or 32
    // LDS row strides (in 4-byte words): fp32 A rows 16 -> 20; bf16 A rows 32 bf16 = 16 words -> 20
    constexpr int AS = BF16 ? 20 : BK + 4;          // both conflict-free for ds_read_b128
    constexpr int A_WORDS = BM * AS;
    // B: fp32 [BK][BN] words; bf16 transposed [BN][BK] bf16 = [BN][16 words] -> stride 20
    // fp32 B rows are padded by 4 words: the two lane-halves of a fragment read rows 8 apart, and
    // 8*(BN+4) = 32 (mod 64) puts them on different bank halves (8*BN = 0 collided: one conflict
    // cycle per MFMA in profiles/r01_pmc_lds_counters.json)
    constexpr int BS = BN + 4;
    constexpr int B_WORDS = BF16 ? BN * AS : BK * BS;
    constexpr int POOL_WORDS = (BM / 16) * BN;
    constexpr int LDS_WORDS = (A_WORDS + B_WORDS) > POOL_WORDS ? (A_WORDS + B_WORDS) : POOL_WORDS;
    __shared__ __attribute__((aligned(16))) unsigned lds[LDS_WORDS];
    unsigned *As = lds;
    unsigned *Bs = lds + A_WORDS;
    __shared__ long long rowpt[ADD ? BM : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, kh = lane >> 5;
    const long long row0 = row_block * BM;
    const int col0 = blockIdx.y * BN;

    // ---- staging assignment -------------------------------------------------------------
    // A: BM x BK elements; thread -> (row = tid/2, half = tid&1): BK/2 consecutive k
    constexpr int A_PER = BK / 2;                    // elements per thread
    const int arow = tid >> 1, ahalf = tid & 1;
    const RowSrc rs = make_row<GATHER>(a, row0 + arow, rows);
    if (ADD && ahalf == 0) rowpt[arow] = rs.valid ? rs.pi : -1;   // read after the K loop's barriers
    const bool a_vec = GATHER ? ((a.c & 3) == 0 && (((uintptr_t)a.feats) & 15) == 0)
                              : ((a.ldx & 3) == 0 && (((uintptr_t)a.x) & 15) == 0);
    // B: BK x BN elements; thread -> k = tid / (BN/BPT), n-offset
    constexpr int B_PER = BK * BN / MLP_THREADS;     // elements per thread
    constexpr int B_TPR = BN / B_PER;                // threads per k-row
    constexpr int BV = (B_PER % 4 == 0) ? 4 : ((B_PER % 2 == 0) ? 2 : 1);   // floats per B load (96-wide tiles: 6 = 3 x 2)
    typedef float bvec_t __attribute__((ext_vector_type(BV)));
    static_assert(B_PER * B_TPR == BN && B_TPR * BK == MLP_THREADS, "B staging must tile the chunk exactly");
    const int bk = tid / B_TPR, bn = (tid % B_TPR) * B_PER;
    const bool b_vec = (a.cout & 3) == 0 && (((uintptr_t)a.w) & 15) == 0;

    f32x4 areg[A_PER / 4];
    float breg[B_PER];

    // Block-uniform fast-path conditions: an interior tile whose chunk lies wholly inside the
    // vector-loadable part of A / B is fetched with unconditional 16-byte loads (no per-element
    // predicates, no branches); everything else takes the general path.
    const bool rows_full = row0 + BM <= rows;
    const int a_klimit = GATHER ? (a.xyz_first ? 0 : a.c) : a.cin;     // A columns [0, a_klimit) are contiguous in memory
    const bool a_fast = rows_full && a_vec;
    const bool b_fast = b_vec && col0 + BN <= a.cout;
    const float *a_base = (GATHER ? a.feats + rs.pi * a.c : a.x + rs.pi * a.ldx) + ahalf * A_PER;
    const float *b_base = a.w + (size_t)bk * a.cout + col0 + bn;

    auto load_chunk = [&](int k0) {
        if (a_fast && k0 + BK <= a_klimit) {
#pragma unroll
            for (int v = 0; v < A_PER / 4; ++v)
                areg[v] = *reinterpret_cast<const f32x4 *>(a_base + k0 + v * 4);
        } else {
#pragma unroll
            for (int v = 0; v < A_PER / 4; ++v)
                areg[v] = load_a4<GATHER>(a, rs, k0 + ahalf * A_PER + v * 4, a_vec);
        }
        if (BV >= 2 && b_fast && k0 + BK <= a.cin) {
            const float *wp = b_base + (size_t)k0 * a.cout;
#pragma unroll
            for (int v = 0; v < B_PER / BV; ++v) {
                const bvec_t t = *reinterpret_cast<const bvec_t *>(wp + v * BV);
#pragma unroll
                for (int e = 0; e < BV; ++e) breg[v * BV + e] = t[e];
            }
            return;
        }
        const int k = k0 + bk;
        const bool kok = k < a.cin;
        const float *wrow = a.w + (size_t)(kok ? k : 0) * a.cout;
#pragma unroll
        for (int v = 0; v < B_PER / BV; ++v) {
            const int n = col0 + bn + v * BV;
            if (BV >= 2 && kok && b_vec && n + BV <= a.cout) {
                const bvec_t t = *reinterpret_cast<const bvec_t *>(wrow + n);
#pragma unroll
                for (int e = 0; e < BV; ++e) breg[v * BV + e] = t[e];
            } else {
#pragma unroll
                for (int e = 0; e < BV; ++e)
                    breg[v * BV + e] = (kok && n + e < a.cout) ? wrow[n + e] : 0.f;
            }
        }
    };
    auto store_chunk = [&]() {
        if (!BF16) {
#pragma unroll
            for (int v = 0; v < A_PER / 4; ++v)
                *reinterpret_cast<f32x4 *>(&As[arow * AS + ahalf * A_PER + v * 4]) = areg[v];
#pragma unroll
            for (int e = 0; e < B_PER; ++e)
                Bs[bk * BS + bn + e] = __builtin_bit_cast(unsigned, breg[e]);
        } else {
            // A: 16 bf16 per thread = 8 words at As[arow][ahalf*8 ..]
#pragma unroll
            for (int v = 0; v < A_PER / 4; ++v) {
                As[arow * AS + ahalf * 8 + v * 2 + 0] = pack_bf16x2(areg[v][0], areg[v][1]);
                As[arow * AS + ahalf * 8 + v * 2 + 1] = pack_bf16x2(areg[v][2], areg[v][3]);
            }
            // B transposed: element (k=bk, n) -> half-word bk of row n
            unsigned short *Bh = reinterpret_cast<unsigned short *>(Bs);
#pragma unroll
            for (int e = 0; e < B_PER; ++e)
                Bh[(bn + e) * (AS * 2) + bk] = (unsigned short)(pack_bf16x2(breg[e], 0.f) & 0xffffu);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (a.cin + BK - 1) / BK;
    load_chunk(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        __syncthreads();                 // previous chunk's reads are done
        store_chunk();
        __syncthreads();
        if (ch + 1 < nchunks) load_chunk((ch + 1) * BK);
        if (!BF16) {
            // A chunk is walked in sub-chunks of 16: MFMA step s of sub-chunk u multiplies physical
            // k = 16u + 8*kh + s, so every output element accumulates its products in the same
            // order whatever the chunk depth (results do not depend on the launch heuristics).
            constexpr int SUBS = BK / 16;
#pragma unroll
            for (int u = 0; u < SUBS; ++u) {
                f32x4 af[TM][2];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const unsigned *p = &As[((wm * TM + i) * 32 + li) * AS + u * 16 + kh * 8];
                    af[i][0] = *reinterpret_cast<const f32x4 *>(p);
                    af[i][1] = *reinterpret_cast<const f32x4 *>(p + 4);
                }
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    float bf[TN];
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bf[j] = __builtin_bit_cast(float, Bs[(u * 16 + kh * 8 + s) * BS + (wn * TN + j) * 32 + li]);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s >> 2][s & 3], bf[j],
                                                                             acc[i][j], 0, 0, 0);
                }
            }
        } else {
            // two k-steps of 16: lane (li, kh) holds k = 16*s + 8*kh + 0..7
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8 af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[i] = *reinterpret_cast<const bf16x8 *>(
                        &As[((wm * TM + i) * 32 + li) * AS + s * 8 + kh * 4]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[j] = *reinterpret_cast<const bf16x8 *>(
                        &Bs[((wn * TN + j) * 32 + li) * AS + s * 8 + kh * 4]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: scale/shift/relu, then store or pool -----------------------------------
    // C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float sc[TN], sh[TN];
    bool cok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + (wn * TN + j) * 32 + li;
        cok[j] = col < a.cout;
        sc[j] = cok[j] ? a.scale[col] : 0.f;
        sh[j] = cok[j] ? a.shift[col] : 0.f;
    }
    if constexpr (ADD) {
        // rowpt[] (written before the K loop by the thread that staged the row) = point row of
        // each of the tile's 128 grouped rows, -1 past the end
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long pt = rowpt[(wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh];
                const float *src = a.add + (pt < 0 ? 0 : pt) * a.ldadd + col0 + li;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float v = (cok[j] && pt >= 0) ? src[(wn * TN + j) * 32] : 0.f;
                    acc[i][j][r] += v;
                }
            }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] * sc[j] + sh[j];
                if (a.relu) v = fmaxf(v, 0.f);
                acc[i][j][r] = v;
            }

    if (a.pool <= 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = col0 + (wn * TN + j) * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = row0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (cok[j] && row < rows) a.out[row * a.ldo + col] = acc[i][j][r];
                }
            }
        return;
    }
    // pooled: 16-row granule maxima -> LDS -> combine pool/16 granules.
    // rows >= a.rows cannot occur inside a group (rows % pool == 0, BM % pool == 0).
    __syncthreads();                     // staging LDS is dead from here
    float *Ps = reinterpret_cast<float *>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 2; ++g) {        // regs 8g..8g+7 = rows 16g..16g+15 of the tile
                float mx = acc[i][j][8 * g];
#pragma unroll
                for (int r = 1; r < 8; ++r) mx = fmaxf(mx, acc[i][j][8 * g + r]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (kh == 0) Ps[(((wm * TM + i) * 2) + g) * BN + (wn * TN + j) * 32 + li] = mx;
            }
    __syncthreads();
    const int gpp = a.pool / 16;                  // granules per pooled group
    const int groups = BM / a.pool;
    for (int e = tid; e < groups * BN; e += MLP_THREADS) {
        const int g = e / BN, cl = e - g * BN;
        const int col = col0 + cl;
        const long long orow = row0 / a.pool + g;
        if (col < a.cout && orow * a.pool < rows) {
            float mx = Ps[(g * gpp) * BN + cl];
            for (int q = 1; q < gpp; ++q) mx = fmaxf(mx, Ps[(g * gpp + q) * BN + cl]);
            a.out[orow * a.ldo + col] = mx;
        }
    }
}

// The kernel proper: one row block per workgroup; with a data-dependent row count (compacted
// rows) the grid is small and every workgroup strides over the row blocks that exist (launching
// the capacity's worth of workgroups only to have them leave costs ~0.1 us each: 100 us at 32768
// rows of capacity, measured).
template <int WM, int WN, int TM, int TN, bool GATHER, bool BF16, int KB, bool ADD = false>
__global__ __launch_bounds__(MLP_THREADS) __attribute__((amdgpu_waves_per_eu((BF16 || GATHER) && TM * TN == 4 ? 3 : 4, 8))) void mlp_kernel(const MlpArgs a)
{
    mlp_tile<WM, WN, TM, TN, GATHER, BF16, KB, ADD>(a, blockIdx.x, a.rows);
}

// Data-dependent row count (compacted rows): a small grid whose workgroups stride over the row
// blocks that exist.  A kernel of its own: the loop state costs the plain kernel registers (spills).
template <int WM, int WN, int TM, int TN, bool GATHER, bool BF16, int KB, bool ADD = false>
__global__ __launch_bounds__(MLP_THREADS) void mlp_kernel_dyn(const MlpArgs a)
{
    const long long r = *a.rows_dev;
    const long long rows = r < a.rows ? r : a.rows;
    for (long long rb = blockIdx.x; rb * (WM * TM * 32) < rows; rb += gridDim.x) {
        mlp_tile<WM, WN, TM, TN, GATHER, BF16, KB, ADD>(a, rb, rows);
        __syncthreads();                              // the next tile reuses the staging LDS
    }
}

// standalone max-pool over consecutive groups of rows (pool sizes the fused epilogue
// does not cover): x (groups*ns, c) -> out (groups, c)
__global__ void max_pool_kernel(long long groups, int ns, int c, const float *__restrict__ x,
                                long long ldx, float *__restrict__ out, long long ldo)
{
    const long long total = groups * c;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long g = e / c;
        const int l = (int)(e - g * c);
        float mx = x[(g * ns) * ldx + l];
        for (int k = 1; k < ns; ++k) mx = fmaxf(mx, x[(g * ns + k) * ldx + l]);
        out[g * ldo + l] = mx;
    }
}

template <bool GATHER, bool BF16>
static int launch_mlp(const MlpArgs &a, hipStream_t st)
{
    const long long rb = (a.rows + BM - 1) / BM;
    if (rb > 0x7fffffffLL) return PRCNN_UNSUPPORTED;
    // tile width: as wide as cout allows, but narrower when the grid would leave most of
    // the 256 CUs (x3 resident workgroups) without work
    // Tile width, from the batch-16 sweeps in profiles/r01_mlp_tile_sweep.txt (3 waves/SIMD build)
    // and r01_mlp_tile_sweep2.txt (current 4 waves/SIMD build): 128 columns
    // amortise the A staging best when K is large or the grid is many rounds deep; 64 columns
    // (4 waves/SIMD instead of 3, twice the workgroups) win on shallow grids; never pad a
    // narrow layer up to a wide tile.
    // diagnostic (PRCNN_BF16_MASK): run classes of bf16 launches in fp32 instead -- 1: 128-wide dense tiles, 2: narrower
    // dense tiles, 4: gather / gather-add prologues, 8: compacted rows
    static const int bf16_mask = getenv("PRCNN_BF16_MASK") ? atoi(getenv("PRCNN_BF16_MASK")) : 15;
    if (BF16 && GATHER && !(bf16_mask & 4)) return launch_mlp<GATHER, false>(a, st);
    if (BF16 && !GATHER && a.rows_dev && !(bf16_mask & 8)) return launch_mlp<GATHER, false>(a, st);
    if (GATHER && a.add) {
        // gather + 3-column product: memory-bound, the narrow tiles (no spills with the gather
        // epilogue) are the right ones whatever cout is
        const int bn = a.cout <= 32 ? 32 : 64;
        dim3 grid((unsigned)(a.rows_dev && rb > DYN_ROW_BLOCKS ? DYN_ROW_BLOCKS : rb), (a.cout + bn - 1) / bn);
        if (a.rows_dev) {
            if (bn == 32)
                hipLaunchKernelGGL((mlp_kernel_dyn<4, 1, 1, 1, GATHER, BF16, BF16 ? 32 : 16, GATHER>), grid, dim3(MLP_THREADS), 0, st, a);
            else
                hipLaunchKernelGGL((mlp_kernel_dyn<4, 1, 1, 2, GATHER, BF16, BF16 ? 32 : 16, GATHER>), grid, dim3(MLP_THREADS), 0, st, a);
        } else if (bn == 32)
            hipLaunchKernelGGL((mlp_kernel<4, 1, 1, 1, GATHER, BF16, BF16 ? 32 : 16, GATHER>), grid, dim3(MLP_THREADS), 0, st, a);
        else
            hipLaunchKernelGGL((mlp_kernel<4, 1, 1, 2, GATHER, BF16, BF16 ? 32 : 16, GATHER>), grid, dim3(MLP_THREADS), 0, st, a);
        PRCNN_LAUNCH_CHECK();
        return PRCNN_OK;
    }
    int bn;
    const long long wgs128 = rb * ((a.cout + 127) / 128);
    if (a.cout <= 32) bn = 32;
    else if (a.cout <= 64) bn = 64;
    else if (a.cout <= 96) bn = 96;   // one 128 x 96 tile per row block: A is staged once (was 3 x 32: 109 -> ? us at K64 N96)
    else if (wgs128 < 256) bn = rb * ((a.cout + 63) / 64) < 256 ? 32 : 64;
    else bn = 128;      // at 4 waves/SIMD the 128-wide tile wins from one workgroup per CU on (sweep 2)
    if (const char *f = getenv("PRCNN_MLP_BN")) bn = atoi(f);   // tuning aid (32 / 64 / 128)
    if (BF16 && !GATHER && !a.rows_dev && !(bf16_mask & (bn == 128 ? 1 : 2))) return launch_mlp<GATHER, false>(a, st);
    // measured (profiles/r01_mlp_tile_sweep.txt): 32-deep fp32 chunks do not pay (the kernel is
    // MFMA-issue bound, not barrier bound), so fp32 stays at 16
    int kb = BF16 ? 32 : 16;
    // except on shallow grids with a long K (few rows: FP levels on 64 / 256 points per cloud), where a
    // workgroup has its SIMDs to itself and the two barriers per chunk are exposed: rows4096 K512
    // N512 31.5 -> 28.9 us, rows1024 K1024 N512 47 -> 32 us with 32-deep chunks
    if (!BF16 && a.cin >= 512 && rb * ((a.cout + bn - 1) / bn) <= 256) kb = 32;
    // and on the 128-wide tile from K = 256 on (round-2 re-measurement, 4 waves/SIMD build: rows65536 K256
    // N256 99.0 -> 92.3 us, rows16384 K512 N512 86.3 -> 83.6 us, rows65536 K608 N256 191 -> 182 us; K128
    // unchanged).  Same accumulation order either way (sub-chunks of 16).
    if (!BF16 && bn == 128 && a.cin >= 256) kb = 32;
    if (const char *f = getenv("PRCNN_MLP_KB")) kb = atoi(f);   // tuning aid (16 / 32)
    if (a.rows_dev) {
        // compacted rows: 64-column tiles, 32-deep chunks (few row blocks exist: more workgroups
        // across the columns, shorter K loops), grid of at most DYN_ROW_BLOCKS row blocks
        const int bnd = a.cout <= 32 ? 32 : 64;
        dim3 gd((unsigned)(rb > DYN_ROW_BLOCKS ? DYN_ROW_BLOCKS : rb), (a.cout + bnd - 1) / bnd);
        if (bnd == 32)
            hipLaunchKernelGGL((mlp_kernel_dyn<4, 1, 1, 1, GATHER, BF16, 32>), gd, dim3(MLP_THREADS), 0, st, a);
        else
            hipLaunchKernelGGL((mlp_kernel_dyn<4, 1, 1, 2, GATHER, BF16, 32>), gd, dim3(MLP_THREADS), 0, st, a);
        PRCNN_LAUNCH_CHECK();
        return PRCNN_OK;
    }
    dim3 grid((unsigned)rb, (a.cout + bn - 1) / bn);
#define PRCNN_MLP_LAUNCH(WM_, WN_, TM_, TN_, KB_) \
    hipLaunchKernelGGL((mlp_kernel<WM_, WN_, TM_, TN_, GATHER, BF16, KB_>), grid, dim3(MLP_THREADS), 0, st, a)
    if (bn == 32) { if (kb == 16) PRCNN_MLP_LAUNCH(4, 1, 1, 1, 16); else PRCNN_MLP_LAUNCH(4, 1, 1, 1, 32); }
    else if (bn == 64) { if (kb == 16) PRCNN_MLP_LAUNCH(4, 1, 1, 2, 16); else PRCNN_MLP_LAUNCH(4, 1, 1, 2, 32); }
    else if (bn == 96) { if (kb == 16) PRCNN_MLP_LAUNCH(4, 1, 1, 3, 16); else PRCNN_MLP_LAUNCH(4, 1, 1, 3, 32); }
    else { if (kb == 16) PRCNN_MLP_LAUNCH(2, 2, 2, 2, 16); else PRCNN_MLP_LAUNCH(2, 2, 2, 2, 32); }
#undef PRCNN_MLP_LAUNCH
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

static int check_common(long long rows, int cin, int cout, const float *w, const float *scale,
                        const float *shift, int pool, int precision, const float *out, long long ldo)
{
    if (rows < 0 || cin <= 0 || cout <= 0) return PRCNN_BAD_SHAPE;
    if (!w || !scale || !shift || !out) return PRCNN_BAD_SHAPE;
    if (precision != 0 && precision != 1) return PRCNN_BAD_ARG;
    if (pool < 1) return PRCNN_BAD_ARG;
    if (ldo < cout) return PRCNN_BAD_SHAPE;
    if (pool > 1) {
        if (pool % 16 != 0 || BM % pool != 0) return PRCNN_UNSUPPORTED;
        if (rows % pool != 0) return PRCNN_BAD_SHAPE;
    }
    return PRCNN_OK;
}

}  // namespace prcnn

extern "C" int prcnn_mlp_layer(int64_t rows, int cin, int cout, const float *x, int64_t ldx,
                               const float *w, const float *scale, const float *shift, int relu,
                               int pool, int precision, float *out, int64_t ldo,
                               prcnn_stream_t stream)
{
    using namespace prcnn;
    int rc = check_common(rows, cin, cout, w, scale, shift, pool, precision, out, ldo);
    if (rc != PRCNN_OK) return rc;
    if (rows == 0) return PRCNN_OK;
    if (!x || ldx < cin) return PRCNN_BAD_SHAPE;
    MlpArgs a = {};
    a.rows = rows; a.cin = cin; a.cout = cout; a.x = x; a.ldx = ldx; a.w = w; a.scale = scale;
    a.shift = shift; a.relu = relu; a.pool = pool; a.out = out; a.ldo = ldo;
    return precision ? launch_mlp<false, true>(a, to_stream(stream))
                     : launch_mlp<false, false>(a, to_stream(stream));
}

extern "C" int prcnn_sa_group_mlp_layer(int b, int n, int m, int nsample, int c, const float *xyz,
                                        const float *feats, const float *new_xyz, const int *idx,
                                        int xyz_first, int cout, const float *w, const float *scale,
                                        const float *shift, int relu, int pool, int precision,
                                        float *out, int64_t ldo, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (b < 0 || n <= 0 || m < 0 || nsample <= 0 || c < 0) return PRCNN_BAD_SHAPE;
    const long long rows = (long long)b * m * nsample;
    int rc = check_common(rows, c + 3, cout, w, scale, shift, pool, precision, out, ldo);
    if (rc != PRCNN_OK) return rc;
    if (rows == 0) return PRCNN_OK;
    if (!xyz || !new_xyz || !idx || (c > 0 && !feats)) return PRCNN_BAD_SHAPE;
    MlpArgs a = {};
    a.rows = rows; a.cin = c + 3; a.cout = cout; a.w = w; a.scale = scale; a.shift = shift;
    a.relu = relu; a.pool = pool; a.out = out; a.ldo = ldo;
    a.n = n; a.m = m; a.ns = nsample; a.c = c; a.xyz_first = xyz_first;
    a.xyz = xyz; a.feats = feats; a.new_xyz = new_xyz; a.idx = idx;
    return precision ? launch_mlp<true, true>(a, to_stream(stream))
                     : launch_mlp<true, false>(a, to_stream(stream));
}

extern "C" int prcnn_sa_group_add_layer(int b, int n, int m, int nsample, const float *xyz,
                                        const float *new_xyz, const int *idx, const float *table,
                                        int64_t ldt, int cout, const float *w_xyz, const float *scale,
                                        const float *shift, int relu, int pool, int precision, float *out,
                                        int64_t ldo, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (b < 0 || n <= 0 || m < 0 || nsample <= 0) return PRCNN_BAD_SHAPE;
    const long long rows = (long long)b * m * nsample;
    int rc = check_common(rows, 3, cout, w_xyz, scale, shift, pool, precision, out, ldo);
    if (rc != PRCNN_OK) return rc;
    if (rows == 0) return PRCNN_OK;
    if (!xyz || !new_xyz || !idx || !table || ldt < cout) return PRCNN_BAD_SHAPE;
    MlpArgs a = {};
    a.rows = rows; a.cin = 3; a.cout = cout; a.w = w_xyz; a.scale = scale; a.shift = shift;
    a.relu = relu; a.pool = pool; a.out = out; a.ldo = ldo;
    a.n = n; a.m = m; a.ns = nsample; a.c = 0; a.xyz_first = 1;
    a.xyz = xyz; a.feats = nullptr; a.new_xyz = new_xyz; a.idx = idx;
    a.add = table; a.ldadd = ldt;
    return precision ? launch_mlp<true, true>(a, to_stream(stream))
                     : launch_mlp<true, false>(a, to_stream(stream));
}

extern "C" int prcnn_max_pool(int64_t groups, int nsample, int c, const float *x, int64_t ldx,
                              float *out, int64_t ldo, prcnn_stream_t stream)
{
    if (groups < 0 || nsample <= 0 || c <= 0 || ldx < c || ldo < c) return PRCNN_BAD_SHAPE;
    if (groups == 0) return PRCNN_OK;
    if (!x || !out) return PRCNN_BAD_SHAPE;
    long long blocks = (groups * c + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(prcnn::max_pool_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       to_stream(stream), (long long)groups, nsample, c, x, (long long)ldx, out,
                       (long long)ldo);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}


// ---- compacted (duplicate-free) grouped rows ------------------------------------------------
namespace prcnn {

// rows of centroid g = its max(cnt,1) distinct slots; off[g] = first row.  Workgroup j owns the
// centroids [1024 j, 1024 j + 1024): it first adds up the row counts of all centroids before its
// chunk (coalesced reads of pts_cnt, 1024 lanes: cheaper than any cross-workgroup hand-off at the
// sizes in use, b*m <= 65536), then scans its own chunk.
__global__ __launch_bounds__(1024) void compact_rows_kernel(long long groups, int n, int m, int ns,
                                                            const int *__restrict__ idx, const int *__restrict__ pts_cnt,
                                                            int *__restrict__ rowpi, int *__restrict__ rowci,
                                                            int *__restrict__ off, int *__restrict__ rows_dev)
{
    __shared__ int wsum[16];
    __shared__ int wpre[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long g0 = (long long)blockIdx.x * 1024;
    if (groups == 0) {
        if (tid == 0) { off[0] = 0; *rows_dev = 0; }
        return;
    }
    auto rows_of = [&](long long g) { const int c = pts_cnt[g]; return c < 1 ? 1 : (c > ns ? ns : c); };
    int pre = 0;
    for (long long g = tid; g < g0; g += 1024) pre += rows_of(g);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) pre += __shfl_xor(pre, d);
    const long long g = g0 + tid;
    const int c = g < groups ? rows_of(g) : 0;
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    if (lane == 0) wpre[wave] = pre;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < 16; ++w) base += wpre[w];
    for (int w = 0; w < wave; ++w) base += wsum[w];
    const int first = base + incl - c;
    if (g < groups) {
        off[g] = first;
        const long long bi = g / m;
        for (int k = 0; k < c; ++k) {
            rowpi[first + k] = (int)(bi * n + idx[g * ns + k]);
            rowci[first + k] = (int)g;
        }
        if (g == groups - 1) { off[groups] = first + c; *rows_dev = first + c; }
    }
}

// out[g][:] = max over rows off[g] .. off[g+1]-1 of x (one wave per group, lanes over columns)
__global__ __launch_bounds__(256) void segment_max_kernel(long long groups, int c, const float *__restrict__ x,
                                                          long long ldx, const int *__restrict__ off,
                                                          float *__restrict__ out, long long ldo)
{
    const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= groups) return;
    const int lane = threadIdx.x & 63;
    const int r0 = off[g], r1 = off[g + 1];
    for (int col = lane; col < c; col += 64) {
        float mx = x[(long long)r0 * ldx + col];
        for (int r = r0 + 1; r < r1; ++r) mx = fmaxf(mx, x[(long long)r * ldx + col]);
        out[g * ldo + col] = mx;
    }
}

}  // namespace prcnn

extern "C" int prcnn_ball_compact_rows(int b, int n, int m, int nsample, const int *idx, const int *pts_cnt,
                                       int *rowpi, int *rowci, int *off, int *rows_dev, prcnn_stream_t stream)
{
    if (b < 0 || n <= 0 || m < 0 || nsample <= 0) return PRCNN_BAD_SHAPE;
    if (!rows_dev || !off) return PRCNN_BAD_SHAPE;
    const long long groups = (long long)b * m;
    if (groups * nsample > 0x7fffffffLL || (long long)b * n > 0x7fffffffLL) return PRCNN_UNSUPPORTED;
    if (groups > 0 && (!idx || !pts_cnt || !rowpi || !rowci)) return PRCNN_BAD_SHAPE;
    hipLaunchKernelGGL(prcnn::compact_rows_kernel, dim3((unsigned)(groups > 0 ? (groups + 1023) / 1024 : 1)), dim3(1024), 0, to_stream(stream), groups, n, m, nsample, idx,
                       pts_cnt, rowpi, rowci, off, rows_dev);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_segment_max(int64_t groups, int c, const float *x, int64_t ldx, const int *off, float *out,
                                 int64_t ldo, prcnn_stream_t stream)
{
    if (groups < 0 || c <= 0 || ldx < c || ldo < c) return PRCNN_BAD_SHAPE;
    if (groups == 0) return PRCNN_OK;
    if (!x || !off || !out) return PRCNN_BAD_SHAPE;
    hipLaunchKernelGGL(prcnn::segment_max_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, to_stream(stream),
                       (long long)groups, c, x, (long long)ldx, off, out, (long long)ldo);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_mlp_layer_n(int64_t max_rows, const int *rows_dev, int cin, int cout, const float *x,
                                 int64_t ldx, const float *w, const float *scale, const float *shift, int relu,
                                 int precision, float *out, int64_t ldo, prcnn_stream_t stream)
{
    using namespace prcnn;
    int rc = check_common(max_rows, cin, cout, w, scale, shift, 1, precision, out, ldo);
    if (rc != PRCNN_OK) return rc;
    if (max_rows == 0) return PRCNN_OK;
    if (!x || ldx < cin || !rows_dev) return PRCNN_BAD_SHAPE;
    MlpArgs a = {};
    a.rows = max_rows; a.cin = cin; a.cout = cout; a.x = x; a.ldx = ldx; a.w = w; a.scale = scale;
    a.shift = shift; a.relu = relu; a.pool = 1; a.out = out; a.ldo = ldo; a.rows_dev = rows_dev;
    return precision ? launch_mlp<false, true>(a, to_stream(stream))
                     : launch_mlp<false, false>(a, to_stream(stream));
}

extern "C" int prcnn_sa_rows_add_layer(int64_t max_rows, const int *rows_dev, const int *rowpi, const int *rowci,
                                       const float *xyz, const float *new_xyz, const float *table, int64_t ldt,
                                       int cout, const float *w_xyz, const float *scale, const float *shift,
                                       int relu, int precision, float *out, int64_t ldo, prcnn_stream_t stream)
{
    using namespace prcnn;
    int rc = check_common(max_rows, 3, cout, w_xyz, scale, shift, 1, precision, out, ldo);
    if (rc != PRCNN_OK) return rc;
    if (max_rows == 0) return PRCNN_OK;
    if (!xyz || !new_xyz || !table || ldt < cout || !rows_dev || !rowpi || !rowci) return PRCNN_BAD_SHAPE;
    MlpArgs a = {};
    a.rows = max_rows; a.cin = 3; a.cout = cout; a.w = w_xyz; a.scale = scale; a.shift = shift;
    a.relu = relu; a.pool = 1; a.out = out; a.ldo = ldo;
    a.n = 1; a.m = 1; a.ns = 1; a.c = 0; a.xyz_first = 1;
    a.xyz = xyz; a.feats = nullptr; a.new_xyz = new_xyz; a.idx = nullptr;
    a.add = table; a.ldadd = ldt;
    a.rows_dev = rows_dev; a.rowpi = rowpi; a.rowci = rowci;
    return precision ? launch_mlp<true, true>(a, to_stream(stream))
                     : launch_mlp<true, false>(a, to_stream(stream));
}
