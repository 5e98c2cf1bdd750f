// Rotated BEV NMS for gfx950 (MI355X).
//
// Replaces nms_rotate_cpu of the reference (utils/nms_rotate.py:39-73: a Python double
// loop around cv2.rotatedRectangleIntersection / convexHull / contourArea).
// Three launches on the caller's stream, float64 arithmetic throughout (MI355X runs
// fp64 VALU at half the fp32 rate, and the reference's boxes/scores are float64):
//   1. rank_kernel     counting sort: rank_i = #{j : s_j > s_i or (s_j == s_i and j > i)}, i.e. the
//                      order of `scores.argsort(kind='stable')[::-1]` -- the reference's line :41
//                      with a stable sort: equal scores come HIGHER index first (NumPy's default
//                      sort is unstable, so the reference itself leaves ties unspecified)
//                      (scores tiled through LDS, j range sliced over workgroups, integer
//                      atomics) + finalize_rank_kernel: order[] and the per-box geometry
//                      record written at its rank -> everything downstream is in score order.
//   2. mask_kernel     all-pairs suppression bitmask, one wave per 64x64 block of the
//                      upper triangle.  Each lane first screens its row against the 64
//                      column boxes with a circumscribed-circle test, the surviving
//                      (row, col) pairs are compacted across the wave, and only those
//                      run the exact convex-polygon clip (Sutherland-Hodgman of one
//                      rectangle against the other in the other's own axis-aligned
//                      frame).  bit(r,c) = iou >= threshold, c > r.
//   3. sweep_kernel    one wave walks the ranks serially: next unsuppressed rank ->
//                      keep, OR its mask row into the LDS-resident `removed` bitmap;
//                      stops at max_output_size (nms_rotate.py:46-47).
// Rectangle convention = OpenCV RotatedRect((b1,b0),(b3,b2),b4) (nms_rotate.py:53,60):
// centre (u,v)=(b1,b0), width W=b3 along (cos t, sin t), height H=b2 along (sin t,-cos t).
#include "common.h"

namespace prcnn {

struct BoxRec {            // 64 bytes, written at the box's rank
    double u, v;           // centre
    double c, s;           // cos / sin of the angle
    double hw, hh;         // half width (b3/2), half height (b2/2)
    double area;           // b2*b3 (nms_rotate.py:55)
    double rad;            // circumscribed radius
};

constexpr int RANK_THREADS = 256;
constexpr int RANK_TILE = 2048;

// Partial ranks: block (x = i-block, y = j-slice, z = scene) counts, for its 256 boxes i,
// the boxes j of its slice that sort before i, and adds the count to rank[i] (zeroed by a
// hipMemsetAsync ahead of the launch).  Integer atomics: the result is order-independent.
// HIGH_FIRST: equal scores sort higher index first (rotated NMS, see the header); otherwise lower
// index first (tf.image.non_max_suppression stand-in, prcnn_nms_aabb_batched).
template <bool HIGH_FIRST>
__global__ __launch_bounds__(RANK_THREADS) void rank_kernel(int n, int slice,
                                                            const double *__restrict__ scores_all,
                                                            int *__restrict__ rank_all)
{
    __shared__ __attribute__((aligned(16))) double ss[RANK_TILE];
    const double *scores = scores_all + (size_t)blockIdx.z * n;
    int *rank = rank_all + (size_t)blockIdx.z * n;
    const int i = blockIdx.x * RANK_THREADS + threadIdx.x;
    const bool ok = i < n;
    const double si = ok ? scores[i] : 0.0;
    const int j0 = blockIdx.y * slice;
    const int j1 = min(n, j0 + slice);
    int cnt = 0;
    for (int base = j0; base < j1; base += RANK_TILE) {
        const int tn = min(RANK_TILE, j1 - base);
        __syncthreads();
        for (int k = threadIdx.x; k < RANK_TILE; k += RANK_THREADS)
            ss[k] = k < tn ? scores[base + k] : __builtin_nan("");   // NaN never sorts before i
        __syncthreads();
        if (ok) {
            // the index comparison only matters for equal scores
            const int tn4 = (tn + 3) & ~3;
            int c = 0;
#pragma unroll 2
            for (int k = 0; k < tn4; k += 4) {
                const double2 a = *reinterpret_cast<const double2 *>(&ss[k]);
                const double2 b = *reinterpret_cast<const double2 *>(&ss[k + 2]);
                const int g = base + k;
                c += (a.x > si || (a.x == si && (HIGH_FIRST ? g > i : g < i))) ? 1 : 0;
                c += (a.y > si || (a.y == si && (HIGH_FIRST ? g + 1 > i : g + 1 < i))) ? 1 : 0;
                c += (b.x > si || (b.x == si && (HIGH_FIRST ? g + 2 > i : g + 2 < i))) ? 1 : 0;
                c += (b.y > si || (b.y == si && (HIGH_FIRST ? g + 3 > i : g + 3 < i))) ? 1 : 0;
            }
            cnt += c;
        }
    }
    if (ok && cnt) atomicAdd(&rank[i], cnt);
}

// order[rank_i] = i and the geometry record of box i at its rank.
__global__ __launch_bounds__(RANK_THREADS) void finalize_rank_kernel(int n, const double *__restrict__ boxes_all,
                                                                     const int *__restrict__ rank_all,
                                                                     int *__restrict__ order_all,
                                                                     BoxRec *__restrict__ rec_all)
{
    const int i = blockIdx.x * RANK_THREADS + threadIdx.x;
    if (i >= n) return;
    const size_t so = (size_t)blockIdx.z * n;
    const int rank = rank_all[so + i];
    order_all[so + rank] = i;
    const double *b = boxes_all + (so + i) * 5;
    const double ang = b[4] * 3.14159265358979323846 / 180.0;
    BoxRec r;
    r.u = b[1]; r.v = b[0];
    sincos(ang, &r.s, &r.c);
    r.hw = 0.5 * b[3]; r.hh = 0.5 * b[2];
    r.area = b[2] * b[3];
    r.rad = sqrt(r.hw * r.hw + r.hh * r.hh);
    rec_all[so + rank] = r;
}

// Clip polygon (px,py)[0..cnt) against the half-plane  sgn*coord <= lim  where coord is x
// (axis 0) or y (axis 1).  Sutherland-Hodgman step; at most cnt+1 output vertices.
__device__ __forceinline__ int clip_axis(const double *px, const double *py, int cnt, int axis,
                                         double sgn, double lim, double *ox, double *oy)
{
    int no = 0;
    for (int i = 0; i < cnt; ++i) {
        const int i2 = (i + 1 == cnt) ? 0 : i + 1;
        const double ax = px[i], ay = py[i], bx = px[i2], by = py[i2];
        const double da = lim - sgn * (axis == 0 ? ax : ay);   // >= 0 inside
        const double db = lim - sgn * (axis == 0 ? bx : by);
        if (da >= 0) { ox[no] = ax; oy[no] = ay; ++no; }
        if ((da >= 0) != (db >= 0)) {
            const double t = da / (da - db);
            ox[no] = ax + t * (bx - ax);
            oy[no] = ay + t * (by - ay);
            ++no;
        }
    }
    return no;
}

// Intersection area of rectangle q with rectangle p, evaluated in p's frame.
__device__ double rect_intersection(const BoxRec &p, const BoxRec &q)
{
    // q's corners (OpenCV RotatedRect::points order) relative to p's centre
    double gx[4], gy[4];
    const double wx = q.c * q.hw, wy = q.s * q.hw;      // half width vector
    const double hx = q.s * q.hh, hy = -q.c * q.hh;     // half height vector
    const double ox = q.u - p.u, oy = q.v - p.v;
    gx[0] = ox - wx - hx; gy[0] = oy - wy - hy;
    gx[1] = ox - wx + hx; gy[1] = oy - wy + hy;
    gx[2] = ox + wx + hx; gy[2] = oy + wy + hy;
    gx[3] = ox + wx - hx; gy[3] = oy + wy - hy;
    double ax[8], ay[8], bx[8], by[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {       // rotate into p's (width, height) axes
        ax[k] = gx[k] * p.c + gy[k] * p.s;
        ay[k] = gx[k] * p.s - gy[k] * p.c;
    }
    int cnt = 4;
    cnt = clip_axis(ax, ay, cnt, 0, 1.0, p.hw, bx, by);
    if (cnt < 3) return 0.0;
    cnt = clip_axis(bx, by, cnt, 0, -1.0, p.hw, ax, ay);
    if (cnt < 3) return 0.0;
    cnt = clip_axis(ax, ay, cnt, 1, 1.0, p.hh, bx, by);
    if (cnt < 3) return 0.0;
    cnt = clip_axis(bx, by, cnt, 1, -1.0, p.hh, ax, ay);
    if (cnt < 3) return 0.0;
    double s = 0.0;
    for (int i = 0; i < cnt; ++i) {
        const int i2 = (i + 1 == cnt) ? 0 : i + 1;
        s += ax[i] * ay[i2] - ax[i2] * ay[i];
    }
    return 0.5 * fabs(s);
}

// One wave per (row block, col block) of the upper triangle; a launch strides over the blocks.
// mode 0: every block.  mode 1 ("head"): only the blocks among the first `head_words` row / column
// blocks, i.e. the pairs of the head_words*64 best-scored boxes.  mode 2 ("tail"): every block
// the head launch did not do, and nothing at all for a frame whose done flag is set (the head
// sweep already found max_output_size boxes: see prcnn_nms_rotate_batched).
__device__ __forceinline__ void tri_decode(long long t, int words, int &rb, int &cb)
{
    // linear block id -> (rb, cb) with cb >= rb: row block rb owns (words - rb) blocks
    // starting at start(rb) = rb*words - rb*(rb-1)/2
    int lo = 0, hi = words - 1;
    while (lo < hi) {                       // largest rb with start(rb) <= t
        const int mid = (lo + hi + 1) >> 1;
        const long long start = (long long)mid * words - (long long)mid * (mid - 1) / 2;
        if (start <= t) lo = mid; else hi = mid - 1;
    }
    rb = lo;
    cb = rb + (int)(t - ((long long)rb * words - (long long)rb * (rb - 1) / 2));
}

__global__ __launch_bounds__(64) void mask_kernel(int n, int words, double thr, double eps,
                                                  const BoxRec *__restrict__ rec_all,
                                                  unsigned long long *__restrict__ mask_all,
                                                  int mode, int head_words, const int *__restrict__ done)
{
    const BoxRec *rec = rec_all + (size_t)blockIdx.z * n;
    unsigned long long *mask = mask_all + (size_t)blockIdx.z * n * words;
    __shared__ BoxRec colb[64];
    __shared__ double colu[64], colv[64], colr[64];
    __shared__ unsigned long long bits[64];
    __shared__ unsigned short pairs[64 * 64];
    if (mode == 2 && done[blockIdx.z]) return;
    const int tw = mode == 1 ? head_words : words;                 // triangle enumerated by this launch
    const long long nblk = (long long)tw * (tw + 1) / 2;
    for (long long t = blockIdx.x; t < nblk; t += gridDim.x) {
    int rb, cb;
    tri_decode(t, tw, rb, cb);
    if (mode == 2 && cb < head_words) continue;                    // done by the head launch
    __syncthreads();                                               // the previous block's LDS reads are over
    {
        const int lane = threadIdx.x;
        const int r = rb * 64 + lane;
        const int cidx = cb * 64 + lane;
        BoxRec cl = {};
        if (cidx < n) cl = rec[cidx];
        colb[lane] = cl;
        colu[lane] = cl.u; colv[lane] = cl.v; colr[lane] = cl.rad;
        bits[lane] = 0ull;
        __syncthreads();
        BoxRec me = {};
        if (r < n) me = rec[r];
        // screen: circumscribed circles, branch-free over the 64 columns (centres and radii as
        // separate LDS arrays: three broadcast reads per column, unrolled); columns past n, at or
        // before the row, and rows past n are masked out afterwards
        unsigned long long cand = 0ull;
#pragma unroll 16
        for (int k = 0; k < 64; ++k) {
            const double dx = colu[k] - me.u, dy = colv[k] - me.v;
            const double rr = colr[k] + me.rad;
            cand |= (unsigned long long)(dx * dx + dy * dy <= rr * rr) << k;
        }
        {
            const int cmax = min(64, n - cb * 64);
            unsigned long long keep = cmax >= 64 ? ~0ull : ((1ull << cmax) - 1ull);
            const int first = r - cb * 64 + 1;                 // first column index past the row
            if (first >= 64) keep = 0ull;
            else if (first > 0) keep &= ~0ull << first;
            if (r >= n) keep = 0ull;
            cand &= keep;
        }
        // compact the candidate pairs of the whole wave into LDS
        const int mine = __builtin_popcountll(cand);
        int incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        int pos = incl - mine;
        const int total = __shfl(incl, 63);
        unsigned long long cc = cand;
        while (cc) {
            const int k = __builtin_ctzll(cc);
            cc &= cc - 1;
            pairs[pos++] = (unsigned short)((lane << 6) | k);
        }
        __syncthreads();
        for (int e = lane; e < total; e += 64) {
            const int pr = pairs[e] >> 6, pk = pairs[e] & 63;
            const BoxRec a = rec[rb * 64 + pr];
            const BoxRec &bq = colb[pk];
            const double ia = rect_intersection(a, bq);
            const double iou = ia * 1.0 / (a.area + bq.area - ia + eps);         // nms_rotate.py:68 (eps 1e-7)
            if (iou >= thr) atomicOr(&bits[pr], 1ull << pk);                       // :70
        }
        __syncthreads();
        if (r < n) mask[(size_t)r * words + cb] = bits[lane];
    }
    }
}

// thr <= 0 suppresses pairs with zero overlap too (0 >= thr): handled by a dense kernel.
__global__ __launch_bounds__(64) void mask_all_kernel(int n, int words,
                                                      unsigned long long *__restrict__ mask_all)
{
    unsigned long long *mask = mask_all + (size_t)blockIdx.z * n * words;
    const long long total = (long long)n * words;
    for (long long e = blockIdx.x * 64ll + threadIdx.x; e < total; e += gridDim.x * 64ll) {
        const int r = (int)(e / words), w = (int)(e - (long long)r * words);
        unsigned long long v = 0ull;
        const int c0 = w * 64;
        for (int k = 0; k < 64; ++k) {
            const int c = c0 + k;
            if (c > r && c < n) v |= 1ull << k;
        }
        mask[e] = v;
    }
}

// Serial sweep by one wave.  `removed` lives in LDS (words <= SWEEP_MAX_WORDS).
constexpr int SWEEP_MAX_WORDS = 8192;   // n <= 524288
constexpr int SWEEP_SPEC = 4;           // mask rows fetched together (words <= 2048, i.e. n <= 131072; else 1); 8 measured no faster
constexpr int SWEEP_SPEC_WORDS = 8192;  // LDS words for them

static int sweep_nspec(int words) { return words <= SWEEP_SPEC_WORDS / SWEEP_SPEC ? SWEEP_SPEC : 1; }
static size_t sweep_lds_bytes(int words) { return (size_t)words * (1 + sweep_nspec(words)) * sizeof(unsigned long long); }

__global__ __launch_bounds__(64) void sweep_kernel(int n, int words, int max_out, int nspec,
                                                   const unsigned long long *__restrict__ mask_all,
                                                   const int *__restrict__ order_all,
                                                   long long *__restrict__ keep_all,
                                                   int *__restrict__ n_keep_all, const int *__restrict__ done)
{
    // done != nullptr ("tail" of the two-stage evaluation): nothing to do for a frame whose head
    // sweep was final; otherwise the whole sweep from rank 0 over the now complete mask.
    if (done && done[blockIdx.x]) return;
    const int stride = words;                             // words per mask row in memory
    const int n_all = n;
    extern __shared__ unsigned long long sweep_lds[];     // removed[words], spec[nspec][words]
    unsigned long long *removed = sweep_lds, *spec = sweep_lds + words;
    const unsigned long long *mask = mask_all + (size_t)blockIdx.x * n_all * stride;
    const int *order = order_all + (size_t)blockIdx.x * n_all;
    long long *keep = keep_all + (size_t)blockIdx.x * max_out;
    int *n_keep = n_keep_all + blockIdx.x;
    const int lane = threadIdx.x;
    for (int w = lane; w < words; w += 64) {
        // bits beyond n are "removed" so the scan never selects them
        const int valid = n - w * 64;
        const unsigned long long v = valid >= 64 ? 0ull : (valid <= 0 ? ~0ull : (~0ull << valid));
        removed[w] = v;
    }
    __syncthreads();
    // Every kept box costs one dependent read of its mask row (global memory, ~1 us): the sweep is
    // latency-bound on it.  So the rows of the next SPEC unsuppressed ranks are fetched together,
    // speculatively (a later one may be suppressed by an earlier one of the same batch: it is then
    // skipped, exactly as the serial order would), and applied in rank order from LDS.
    auto first_open = [&](int from) -> int {     // first zero bit of `removed` at position >= from
        for (int w0 = from >> 6; w0 < words; w0 += 64) {
            const int w = w0 + lane;
            unsigned long long z = w < words ? ~removed[w] : 0ull;
            if (w == (from >> 6)) z &= ~0ull << (from & 63);
            const unsigned long long has = __ballot(z != 0ull);
            if (has) {
                const int fl = __builtin_ctzll(has);
                return (w0 + fl) * 64 + __builtin_ctzll(__shfl(z, fl));
            }
        }
        return -1;
    };
    int nk = 0;
    long long mykeep = 0;
    int cur = 0;                                 // first rank not yet examined
    while (nk < max_out && cur < n) {
        // (fully unrolled with compile-time indices: cands stays in registers)
        int cands[SWEEP_SPEC];
        int nc = 0, last = cur - 1;
        bool more = true;
#pragma unroll
        for (int q = 0; q < SWEEP_SPEC; ++q) {
            cands[q] = -1;
            if (more && q < nspec) {
                const int f = first_open(last + 1);
                if (f < 0 || f >= n) more = false;
                else { cands[q] = f; last = f; nc = q + 1; }
            }
        }
        if (nc == 0) break;
        // all loads of a slab (4 words per lane and row) are issued before the first is used
        int ord[SWEEP_SPEC];                     // original indices of the candidates (uniform loads)
#pragma unroll
        for (int q = 0; q < SWEEP_SPEC; ++q) ord[q] = q < nc ? order[cands[q]] : 0;
        for (int base = (cands[0] >> 6) & ~63; base < words; base += 256) {
            unsigned long long v[SWEEP_SPEC][4];
#pragma unroll
            for (int q = 0; q < SWEEP_SPEC; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int w = base + j * 64 + lane;
                    v[q][j] = (q < nc && w < words) ? mask[(size_t)cands[q] * stride + w] : 0ull;
                }
#pragma unroll
            for (int q = 0; q < SWEEP_SPEC; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int w = base + j * 64 + lane;
                    if (q < nc && w < words) spec[q * words + w] = v[q][j];
                }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SWEEP_SPEC; ++q) {
            if (q < nc && nk < max_out) {
                const int f = cands[q];
                const bool gone = q > 0 && ((removed[f >> 6] >> (f & 63)) & 1ull);   // suppressed inside the batch
                if (!gone) {
                    // kept indices collect in registers (lane nk % 64) and leave 64 at a time: a
                    // store per kept box would be waited for at the barrier below
                    if (lane == (nk & 63)) mykeep = ord[q];
                    ++nk;
                    if ((nk & 63) == 0) keep[nk - 64 + lane] = mykeep;
                    // removed |= mask row of f (only words >= f/64 matter)
                    for (int w = (f >> 6) + lane; w < words; w += 64) removed[w] |= spec[q * words + w];
                    __syncthreads();
                }
            }
        }
        cur = last + 1;
    }
    if (lane < (nk & 63)) keep[(nk & ~63) + lane] = mykeep;
    if (lane == 0) *n_keep = nk;
}

// The head sweep (mode 1 of the two-stage evaluation): the pair bits of the HEAD_RANKS best-scored
// boxes and their original indices are first copied into LDS (32 KB + 2 KB, all loads in flight
// together), the `removed` bitmap is one register per lane, and the serial walk then costs a
// ballot, an LDS row read and an OR per kept box -- no global-memory latency on the chain
// (the general sweep pays ~0.7 us per kept box for its mask row).  One wave per frame.
constexpr int HEAD_WORDS = 8;
constexpr int HEAD_RANKS = HEAD_WORDS * 64;

__global__ __launch_bounds__(HEAD_RANKS) void sweep_head_kernel(int n_all, int stride, int max_out,
                                                        const unsigned long long *__restrict__ mask_all,
                                                        const int *__restrict__ order_all,
                                                        long long *__restrict__ keep_all,
                                                        int *__restrict__ n_keep_all, int *__restrict__ done)
{
    __shared__ unsigned long long hm[HEAD_RANKS * HEAD_WORDS];
    __shared__ int hord[HEAD_RANKS];
    const unsigned long long *mask = mask_all + (size_t)blockIdx.x * n_all * stride;
    const int *order = order_all + (size_t)blockIdx.x * n_all;
    long long *keep = keep_all + (size_t)blockIdx.x * max_out;
    const int lane = threadIdx.x & 63;
    const int n = n_all < HEAD_RANKS ? n_all : HEAD_RANKS;
    const int wl = (n + 63) >> 6;
    {
        // 512 threads, 8 mask words each, every load issued before the first is stored
        unsigned long long v[HEAD_WORDS];
#pragma unroll
        for (int i = 0; i < HEAD_WORDS; ++i) {
            const int e = threadIdx.x + i * HEAD_RANKS;
            const int r = e >> 3, w = e & 7;
            // words left of the diagonal were never written (and are never needed)
            v[i] = (r < n && w < wl && w >= (r >> 6)) ? mask[(size_t)r * stride + w] : 0ull;
        }
#pragma unroll
        for (int i = 0; i < HEAD_WORDS; ++i) hm[threadIdx.x + i * HEAD_RANKS] = v[i];
        if ((int)threadIdx.x < n) hord[threadIdx.x] = order[threadIdx.x];
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;                       // the walk itself is one wave's work
    // lane w < wl holds word w of `removed`; bits beyond n are set so the scan never selects them
    unsigned long long rem = ~0ull;
    if (lane < wl) {
        const int valid = n - lane * 64;
        rem = valid >= 64 ? 0ull : (~0ull << valid);
    }
    int nk = 0, cur = 0;
    long long mykeep = 0;
    while (nk < max_out) {
        unsigned long long z = ~rem;
        if (lane == (cur >> 6)) z &= ~0ull << (cur & 63);
        if (lane < (cur >> 6)) z = 0ull;
        const unsigned long long has = __ballot(z != 0ull);
        if (!has) break;
        const int fl = __builtin_ctzll(has);
        const int f = fl * 64 + __builtin_ctzll(__shfl(z, fl));
        if (lane == (nk & 63)) mykeep = f;                 // rank now, original index when stored
        ++nk;
        if ((nk & 63) == 0) keep[nk - 64 + lane] = hord[mykeep];
        if (lane < HEAD_WORDS) rem |= hm[f * HEAD_WORDS + lane];
        cur = f + 1;
        if (cur >= n) break;
    }
    if (lane < (nk & 63)) keep[(nk & ~63) + lane] = hord[mykeep];
    if (lane == 0) {
        n_keep_all[blockIdx.x] = nk;
        done[blockIdx.x] = (nk >= max_out || n >= n_all) ? 1 : 0;
    }
}

static int launch_sweep(int batch, int n, int words, int max_out, const unsigned long long *mask, const int *order,
                        long long *keep, int *n_keep, hipStream_t st, const int *done = nullptr)
{
    const size_t lds = sweep_lds_bytes(words);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)sweep_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
            hipSuccess)
            return PRCNN_HIP_ERROR;
    }
    hipLaunchKernelGGL(sweep_kernel, dim3(batch), dim3(64), lds, st, n, words, max_out, sweep_nspec(words), mask,
                       order, keep, n_keep, done);
    return PRCNN_OK;
}

struct NmsLayout {
    size_t rank_off, order_off, rec_off, mask_off, done_off, total;
    int words;
};
static constexpr NmsLayout nms_layout(int batch, int n)
{
    NmsLayout L{};
    L.words = (n + 63) / 64;
    const size_t B = (size_t)batch;
    size_t off = 0;
    L.rank_off = off; off += (B * n * sizeof(int) + 255) & ~(size_t)255;
    L.order_off = off; off += (B * n * sizeof(int) + 255) & ~(size_t)255;
    L.rec_off = off; off += (B * n * sizeof(BoxRec) + 255) & ~(size_t)255;
    L.mask_off = off; off += (B * n * L.words * sizeof(unsigned long long) + 255) & ~(size_t)255;
    L.done_off = off; off += (B * sizeof(int) + 255) & ~(size_t)255;
    L.total = off + 256;
    return L;
}


// ---- axis-aligned NMS with tf.image.non_max_suppression semantics (models/rpn.py:236) ----------
// TensorFlow's NonMaxSuppression (third-party, un-vendored; TF 1.x core/kernels/
// non_max_suppression_op.cc): candidates in decreasing score order; a candidate is dropped when
// its IoU with an already selected box is > iou_threshold; boxes are (y1,x1,y2,x2) in any corner
// order; a box of non-positive area never suppresses and is never suppressed; all in float.
// Same rank / bitmask / sweep structure as the rotated NMS; the record is the normalised box.
struct AabbRec { float ymin, xmin, ymax, xmax, area, pad0, pad1, pad2; };

__global__ __launch_bounds__(RANK_THREADS) void aabb_finalize_kernel(int n, const float *__restrict__ boxes_all,
                                                                     const int *__restrict__ rank_all,
                                                                     int *__restrict__ order_all,
                                                                     AabbRec *__restrict__ rec_all)
{
    const int i = blockIdx.x * RANK_THREADS + threadIdx.x;
    if (i >= n) return;
    const size_t so = (size_t)blockIdx.z * n;
    const int rank = rank_all[so + i];
    order_all[so + rank] = i;
    const float *b = boxes_all + (so + i) * 4;
    AabbRec r;
    r.ymin = fminf(b[0], b[2]); r.xmin = fminf(b[1], b[3]);
    r.ymax = fmaxf(b[0], b[2]); r.xmax = fmaxf(b[1], b[3]);
    r.area = (r.ymax - r.ymin) * (r.xmax - r.xmin);
    r.pad0 = r.pad1 = r.pad2 = 0.f;
    rec_all[so + rank] = r;
}

__global__ __launch_bounds__(64) void aabb_mask_kernel(int n, int words, float thr,
                                                       const AabbRec *__restrict__ rec_all,
                                                       unsigned long long *__restrict__ mask_all)
{
    const AabbRec *rec = rec_all + (size_t)blockIdx.z * n;
    unsigned long long *mask = mask_all + (size_t)blockIdx.z * n * words;
    __shared__ AabbRec colb[64];
    int rb, cb;
    {
        const long long t = blockIdx.x;
        int lo = 0, hi = words - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            const long long start = (long long)mid * words - (long long)mid * (mid - 1) / 2;
            if (start <= t) lo = mid; else hi = mid - 1;
        }
        rb = lo;
        cb = rb + (int)(t - ((long long)rb * words - (long long)rb * (rb - 1) / 2));
    }
    const int lane = threadIdx.x;
    const int r = rb * 64 + lane;
    if (cb * 64 + lane < n) colb[lane] = rec[cb * 64 + lane];
    __syncthreads();
    if (r >= n) return;
    const AabbRec me = rec[r];
    unsigned long long bits = 0ull;
    const int cmax = min(64, n - cb * 64);
    for (int k = 0; k < cmax; ++k) {
        if (cb * 64 + k <= r) continue;
        const AabbRec o = colb[k];
        if (me.area <= 0.f || o.area <= 0.f) continue;
        const float iy = fmaxf(fminf(me.ymax, o.ymax) - fmaxf(me.ymin, o.ymin), 0.f);
        const float ix = fmaxf(fminf(me.xmax, o.xmax) - fmaxf(me.xmin, o.xmin), 0.f);
        const float inter = iy * ix;
        const float iou = inter / (me.area + o.area - inter);
        if (iou > thr) bits |= 1ull << k;
    }
    mask[(size_t)r * words + cb] = bits;
}

}  // namespace prcnn

extern "C" int prcnn_nms_aabb_batched(int batch, int n, const float *boxes, const double *scores,
                                      float iou_threshold, int max_output_size, int64_t *keep, int *n_keep,
                                      void *workspace, size_t workspace_bytes, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (n < 0 || batch < 0) return PRCNN_BAD_SHAPE;
    if (batch == 0) return PRCNN_OK;
    if (!n_keep) return PRCNN_BAD_SHAPE;
    if (batch > 65535) return PRCNN_UNSUPPORTED;
    hipStream_t st = to_stream(stream);
    if (n == 0 || max_output_size <= 0) {
        if (hipMemsetAsync(n_keep, 0, sizeof(int) * (size_t)batch, st) != hipSuccess) return PRCNN_HIP_ERROR;
        return PRCNN_OK;
    }
    if (!boxes || !scores || !keep) return PRCNN_BAD_SHAPE;
    const NmsLayout L = nms_layout(batch, n);
    if (L.words > SWEEP_MAX_WORDS) return PRCNN_UNSUPPORTED;
    if (!workspace || workspace_bytes < L.total) return PRCNN_WORKSPACE;
    if (((uintptr_t)workspace & 255) != 0) return PRCNN_WORKSPACE;
    char *ws = (char *)workspace;
    int *rank = (int *)(ws + L.rank_off);
    int *order = (int *)(ws + L.order_off);
    AabbRec *rec = (AabbRec *)(ws + L.rec_off);          // 32-byte records in the 64-byte slots
    unsigned long long *mask = (unsigned long long *)(ws + L.mask_off);
    if (hipMemsetAsync(rank, 0, sizeof(int) * (size_t)batch * n, st) != hipSuccess) return PRCNN_HIP_ERROR;
    const int iblocks = (n + RANK_THREADS - 1) / RANK_THREADS;
    int slices = (1024 + iblocks * batch - 1) / (iblocks * batch);
    const int max_slices = (n + 511) / 512;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    const int slice = (((n + slices - 1) / slices) + 3) & ~3;
    slices = (n + slice - 1) / slice;
    hipLaunchKernelGGL(rank_kernel<false>, dim3(iblocks, slices, batch), dim3(RANK_THREADS), 0, st, n, slice, scores, rank);
    hipLaunchKernelGGL(aabb_finalize_kernel, dim3(iblocks, 1, batch), dim3(RANK_THREADS), 0, st, n, boxes, rank,
                       order, rec);
    const long long nblk = (long long)L.words * (L.words + 1) / 2;
    hipLaunchKernelGGL(aabb_mask_kernel, dim3((unsigned)nblk, 1, batch), dim3(64), 0, st, n, L.words,
                       iou_threshold, rec, mask);
    if (launch_sweep(batch, n, L.words, max_output_size, mask, order, (long long *)keep, n_keep, st) != PRCNN_OK)
        return PRCNN_HIP_ERROR;
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

// ---- final detection de-dup: nms_on_bev / group_overlaps (test/test_frustum.py:185-218) ---------
// Greedy grouping in index order: the first remaining detection collects every later remaining
// detection whose rotated-BEV IoU with it is >= iou_threshold (get_iou :185-189: shapely polygon
// intersection over union, no epsilon); the highest score of a group survives (stable sort,
// :215: the earliest of equal scores).  BEV rectangle = corners 0..3 of compute_box_3d
// (kitti/kitti_util.py:322-339): centre (x,z), length l along (cos ry, -sin ry), width w.
namespace prcnn {

__global__ __launch_bounds__(RANK_THREADS) void bev_rec_kernel(int n, const int *__restrict__ counts,
                                                               const double *__restrict__ boxes_all,
                                                               BoxRec *__restrict__ rec_all)
{
    const int i = blockIdx.x * RANK_THREADS + threadIdx.x;
    if (i >= n) return;
    const size_t so = (size_t)blockIdx.z * n;
    const int cnt = counts ? counts[blockIdx.z] : n;
    const double *b = boxes_all + (so + i) * 5;
    BoxRec r;
    r.u = i < cnt ? b[0] : __builtin_nan("");        // NaN centre: never passes the circle screen
    r.v = b[1];
    double sn, cs;
    sincos(b[4], &sn, &cs);
    r.c = cs; r.s = -sn;
    r.hw = 0.5 * b[2]; r.hh = 0.5 * b[3];
    r.area = b[2] * b[3];
    r.rad = sqrt(r.hw * r.hw + r.hh * r.hh);
    rec_all[so + i] = r;
}

constexpr int GROUP_MAX_WORDS = 1024;   // n <= 65536 detections per frame

__global__ __launch_bounds__(64) void bev_group_kernel(int n, int words, const int *__restrict__ counts,
                                                       const unsigned long long *__restrict__ mask_all,
                                                       const double *__restrict__ scores_all,
                                                       int *__restrict__ group_all, int *__restrict__ keep_all,
                                                       int *__restrict__ n_keep_all)
{
    __shared__ unsigned long long removed[GROUP_MAX_WORDS];
    const size_t so = (size_t)blockIdx.x * n;
    const unsigned long long *mask = mask_all + so * words;
    const double *scores = scores_all + so;
    int *group = group_all + so, *keep = keep_all + so;
    const int lane = threadIdx.x;
    const int cnt = min(n, max(0, counts ? counts[blockIdx.x] : n));
    for (int w = lane; w < words; w += 64) {
        const int valid = cnt - w * 64;
        removed[w] = valid >= 64 ? 0ull : (valid <= 0 ? ~0ull : (~0ull << valid));
    }
    for (int i = cnt + lane; i < n; i += 64) group[i] = -1;
    __syncthreads();
    int ng = 0, cur = 0;
    while (cur < cnt) {
        int found = -1;
        for (int w0 = cur >> 6; w0 < words && found < 0; w0 += 64) {
            const int w = w0 + lane;
            unsigned long long z = w < words ? ~removed[w] : 0ull;
            if (w == (cur >> 6)) z &= ~0ull << (cur & 63);
            const unsigned long long has = __ballot(z != 0ull);
            if (has) {
                const int fl = __builtin_ctzll(has);
                found = (w0 + fl) * 64 + __builtin_ctzll(__shfl(z, fl));
            }
        }
        if (found < 0 || found >= cnt) break;
        // members = {found} + remaining later detections overlapping it; best = max score, earliest
        const unsigned long long *row = mask + (size_t)found * words;
        double bs = 0.0;
        int bi = 0x7fffffff;
        for (int w = (found >> 6) + lane; w < words; w += 64) {
            unsigned long long mem = row[w] & ~removed[w];
            if (w == (found >> 6)) mem |= 1ull << (found & 63);
            removed[w] |= mem;
            while (mem) {
                const int i = w * 64 + __builtin_ctzll(mem);
                mem &= mem - 1;
                group[i] = ng;
                const double s = scores[i];
                if (bi == 0x7fffffff || s > bs) { bs = s; bi = i; }      // ascending i: first maximum
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double os = __shfl_xor(bs, off);
            const int oi = __shfl_xor(bi, off);
            const bool take = oi != 0x7fffffff && (bi == 0x7fffffff || os > bs || (os == bs && oi < bi));
            if (take) { bs = os; bi = oi; }
        }
        // a group whose scores are all NaN keeps its first member, like Python's sort of NaNs may
        if (lane == 0) keep[ng] = bi;
        ++ng;
        __syncthreads();
        cur = found + 1;
    }
    for (int i = ng + lane; i < n; i += 64) keep[i] = -1;
    if (lane == 0) n_keep_all[blockIdx.x] = ng;
}

}  // namespace prcnn

extern "C" int prcnn_bev_group_keep(int frames, int n, const int *counts, const double *boxes,
                                    const double *scores, double iou_threshold, int *group_id, int *keep,
                                    int *n_keep, void *workspace, size_t workspace_bytes, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (n < 0 || frames < 0) return PRCNN_BAD_SHAPE;
    if (frames == 0) return PRCNN_OK;
    if (!n_keep) return PRCNN_BAD_SHAPE;
    if (frames > 65535) return PRCNN_UNSUPPORTED;
    if (!(iou_threshold > 0.0)) return PRCNN_UNSUPPORTED;      // 0 / NaN: every pair, not a de-dup
    hipStream_t st = to_stream(stream);
    if (n == 0) {
        if (hipMemsetAsync(n_keep, 0, sizeof(int) * (size_t)frames, st) != hipSuccess) return PRCNN_HIP_ERROR;
        return PRCNN_OK;
    }
    if (!boxes || !scores || !group_id || !keep) return PRCNN_BAD_SHAPE;
    const NmsLayout L = nms_layout(frames, n);
    if (L.words > GROUP_MAX_WORDS) return PRCNN_UNSUPPORTED;
    if (!workspace || workspace_bytes < L.total) return PRCNN_WORKSPACE;
    if (((uintptr_t)workspace & 255) != 0) return PRCNN_WORKSPACE;
    char *ws = (char *)workspace;
    BoxRec *rec = (BoxRec *)(ws + L.rec_off);
    unsigned long long *mask = (unsigned long long *)(ws + L.mask_off);
    const int iblocks = (n + RANK_THREADS - 1) / RANK_THREADS;
    hipLaunchKernelGGL(bev_rec_kernel, dim3(iblocks, 1, frames), dim3(RANK_THREADS), 0, st, n, counts, boxes, rec);
    const long long nblk = (long long)L.words * (L.words + 1) / 2;
    hipLaunchKernelGGL(mask_kernel, dim3((unsigned)nblk, 1, frames), dim3(64), 0, st, n, L.words, iou_threshold,
                       0.0, rec, mask, 0, 0, nullptr);
    hipLaunchKernelGGL(bev_group_kernel, dim3(frames), dim3(64), 0, st, n, L.words, counts, mask, scores,
                       group_id, keep, n_keep);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" size_t prcnn_nms_rotate_batched_workspace_bytes(int batch, int n)
{
    if (n <= 0 || batch <= 0) return 256;
    return prcnn::nms_layout(batch, n).total;
}

extern "C" size_t prcnn_nms_rotate_workspace_bytes(int n)
{
    return prcnn_nms_rotate_batched_workspace_bytes(1, n);
}

// The launches of prcnn_nms_rotate_batched on a workspace `ws` of nms_layout(batch, n).total bytes
// (256-byte aligned); shared by the caller-workspace and the static-scratch entry points.
static int nms_rotate_run(int batch, int n, const double *boxes, const double *scores, double iou_threshold,
                          int max_output_size, int64_t *keep, int *n_keep, void *workspace,
                          size_t workspace_bytes, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (n < 0 || batch < 0) return PRCNN_BAD_SHAPE;
    if (batch == 0) return PRCNN_OK;
    if (!n_keep) return PRCNN_BAD_SHAPE;
    if (batch > 65535) return PRCNN_UNSUPPORTED;
    hipStream_t st = to_stream(stream);
    if (n == 0 || max_output_size <= 0) {
        if (hipMemsetAsync(n_keep, 0, sizeof(int) * (size_t)batch, st) != hipSuccess) return PRCNN_HIP_ERROR;
        return PRCNN_OK;
    }
    if (!boxes || !scores || !keep) return PRCNN_BAD_SHAPE;
    const NmsLayout L = nms_layout(batch, n);
    if (L.words > SWEEP_MAX_WORDS) return PRCNN_UNSUPPORTED;
    if (!workspace || workspace_bytes < L.total) return PRCNN_WORKSPACE;
    if (((uintptr_t)workspace & 255) != 0) return PRCNN_WORKSPACE;
    char *ws = (char *)workspace;
    int *rank = (int *)(ws + L.rank_off);
    int *order = (int *)(ws + L.order_off);
    BoxRec *rec = (BoxRec *)(ws + L.rec_off);
    unsigned long long *mask = (unsigned long long *)(ws + L.mask_off);

    if (hipMemsetAsync(rank, 0, sizeof(int) * (size_t)batch * n, st) != hipSuccess) return PRCNN_HIP_ERROR;
    const int iblocks = (n + RANK_THREADS - 1) / RANK_THREADS;
    // enough j-slices to put >= ~1024 workgroups on the chip, each at least one tile long
    int slices = (1024 + iblocks * batch - 1) / (iblocks * batch);
    const int max_slices = (n + 511) / 512;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    const int slice = (((n + slices - 1) / slices) + 3) & ~3;
    slices = (n + slice - 1) / slice;
    hipLaunchKernelGGL(rank_kernel<true>, dim3(iblocks, slices, batch), dim3(RANK_THREADS), 0, st, n, slice,
                       scores, rank);
    PRCNN_LAUNCH_CHECK();
    hipLaunchKernelGGL(finalize_rank_kernel, dim3(iblocks, 1, batch), dim3(RANK_THREADS), 0, st, n, boxes,
                       rank, order, rec);
    PRCNN_LAUNCH_CHECK();
    if (iou_threshold > 0.0) {
        // Only the upper triangle (incl. the diagonal blocks) is ever read by the sweep -- and the
        // sweep stops at max_output_size kept boxes, usually long before the last rank: with the
        // reference's settings (IoU 0.8 -> 100 of 2048, or of 9000) it ends within the first ~120
        // ranks and ~99.9 % of the pair tests would never be looked at.  So the pairs among the
        // HEAD_RANKS best-scored boxes are evaluated first ("head": 36 blocks per frame) and swept;
        // if that sweep keeps max_output_size boxes (or the frame has no more boxes) its result is
        // the full sweep's result -- the serial sweep never reads a column beyond the rank it stops
        // at -- and the frame's done flag makes the tail launches (all remaining blocks, then the
        // whole sweep over the complete mask) leave at once.  Same keep list in every case.
        int *done = (int *)(ws + L.done_off);
        const int hw = L.words < HEAD_WORDS ? L.words : HEAD_WORDS;
        const long long hblk = (long long)hw * (hw + 1) / 2;
        hipLaunchKernelGGL(mask_kernel, dim3((unsigned)hblk, 1, batch), dim3(64), 0, st, n, L.words,
                           iou_threshold, 0.0000001, rec, mask, 1, hw, nullptr);
        PRCNN_LAUNCH_CHECK();
        hipLaunchKernelGGL(sweep_head_kernel, dim3(batch), dim3(HEAD_RANKS), 0, st, n, L.words, max_output_size, mask,
                           order, (long long *)keep, n_keep, done);
        PRCNN_LAUNCH_CHECK();
        if (L.words > hw) {
            const long long nblk = (long long)L.words * (L.words + 1) / 2;
            // a bounded grid that strides over the blocks: a frame that is done costs its
            // workgroups one flag read each
            const long long per_frame = batch >= 8 ? 256 : 2048 / batch;
            const unsigned gx = (unsigned)(nblk < per_frame ? nblk : per_frame);
            hipLaunchKernelGGL(mask_kernel, dim3(gx, 1, batch), dim3(64), 0, st, n, L.words, iou_threshold,
                               0.0000001, rec, mask, 2, hw, done);
            PRCNN_LAUNCH_CHECK();
            if (launch_sweep(batch, n, L.words, max_output_size, mask, order, (long long *)keep, n_keep, st,
                             done) != PRCNN_OK)
                return PRCNN_HIP_ERROR;
        }
    } else {
        hipLaunchKernelGGL(mask_all_kernel, dim3(2048, 1, batch), dim3(64), 0, st, n, L.words, mask);
        PRCNN_LAUNCH_CHECK();
        if (launch_sweep(batch, n, L.words, max_output_size, mask, order, (long long *)keep, n_keep, st) != PRCNN_OK)
            return PRCNN_HIP_ERROR;
    }
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}

extern "C" int prcnn_nms_rotate_batched(int batch, int n, const double *boxes, const double *scores,
                                        double iou_threshold, int max_output_size, int64_t *keep,
                                        int *n_keep, void *workspace, size_t workspace_bytes,
                                        prcnn_stream_t stream)
{
    return nms_rotate_run(batch, n, boxes, scores, iou_threshold, max_output_size, keep, n_keep, workspace,
                          workspace_bytes, stream);
}

extern "C" int prcnn_nms_rotate(int n, const double *boxes, const double *scores,
                                double iou_threshold, int max_output_size, int64_t *keep,
                                int *n_keep, void *workspace, size_t workspace_bytes,
                                prcnn_stream_t stream)
{
    return prcnn_nms_rotate_batched(1, n, boxes, scores, iou_threshold, max_output_size, keep, n_keep,
                                    workspace, workspace_bytes, stream);
}

// ---- allocation-free entry point ------------------------------------------------------------
// prcnn_nms_rotate for callers that cannot allocate (code under HIP graph capture, tight serving
// loops, harnesses that forbid hipMalloc): the workspace (rank, order, box records, suppression
// bitmask) is a file-scope device array instead of a caller buffer.  Each GPU the library runs on
// holds its own copy of the array (PRCNN_NMS_NOALLOC_MAX_N boxes: ~11.3 MB) until the process
// ends.  In host code the array's NAME is only a stand-in: the device address comes from
// hipGetSymbolAddress, once per device, with that device current, and only that pointer is ever
// handed to kernels and memsets.  Calls on one GPU share the array: two calls that overlap on one
// GPU (different streams or threads) race; calls on one stream are ordered and every call clears
// what it needs (the rank counters), so stale contents of an earlier, larger call are never read.
namespace prcnn {
constexpr int NOALLOC_MAX_N = PRCNN_NMS_NOALLOC_MAX_N;
constexpr size_t NOALLOC_BYTES = nms_layout(1, NOALLOC_MAX_N).total;
constexpr int NOALLOC_MAX_DEVICES = 64;
}  // namespace prcnn

static __device__ __align__(256) unsigned char scratch[prcnn::NOALLOC_BYTES];

#include <mutex>
static std::mutex g_scratch_mu;
static void *g_scratch_ptr[prcnn::NOALLOC_MAX_DEVICES] = {};

// device address of `scratch` on the CURRENT device (resolved on the first call per device)
static int noalloc_scratch(void **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        prcnn_note_hip_error((int)e);
        return PRCNN_HIP_ERROR;
    }
    if (dev < 0 || dev >= prcnn::NOALLOC_MAX_DEVICES) return PRCNN_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) {
            prcnn_note_hip_error((int)e);
            return PRCNN_HIP_ERROR;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return PRCNN_OK;
}

extern "C" int prcnn_nms_rotate_noalloc(int n, const double *boxes, const double *scores,
                                        double iou_threshold, int max_output_size, int64_t *keep,
                                        int *n_keep, prcnn_stream_t stream)
{
    if (n < 0) return PRCNN_BAD_SHAPE;
    if (n > prcnn::NOALLOC_MAX_N) return PRCNN_UNSUPPORTED;     // beyond the static scratch: use prcnn_nms_rotate
    void *ws = nullptr;
    const int rc = noalloc_scratch(&ws);
    if (rc != PRCNN_OK) return rc;
    return nms_rotate_run(1, n, boxes, scores, iou_threshold, max_output_size, keep, n_keep, ws,
                          prcnn::NOALLOC_BYTES, stream);
}
