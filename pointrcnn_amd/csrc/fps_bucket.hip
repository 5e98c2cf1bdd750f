// Farthest point sampling with a small footprint: the sampler that SHARES its CU (additive; same
// result as prcnn_farthest_point_sample, tf_sampling_g.cu:105-181).
//
// The register-resident samplers of fps.hip hold a whole cloud in one CU's VGPRs (16 waves x 94
// registers + 100 KB of LDS for the in-workgroup sort at 16 384 points): fastest for one cloud, but
// the CU is lost to every other kernel for the 2.9 ms the chain takes, which is what a pipelined
// caller pays for sampling (bench.py: 58 of ~350 CU-ms per batch).  Here the cloud stays
// in global memory (L2-resident: 320 KB per cloud), sorted into buckets of 128 neighbouring points
// by a short separate launch; the sampling workgroup is 4 waves that keep only the 128 bucket boxes
// and bucket maxima in registers (one bucket per lane, ~50 VGPRs, 24 bytes of LDS) and load, per
// iteration, just the buckets whose box comes closer to the new sample than their largest running
// distance -- a handful.  An iteration costs one more L2 round trip than the register-resident
// kernel, but four waves with ~50 registers fit beside any other kernel's waves, so the chain no
// longer owns a CU.
//
// Bit-exactness: same arithmetic and the same (max value, smallest k mod 512, smallest k) winner
// on ORIGINAL indices as fps_kernel / fps_pruned_kernel (fps.hip); the bound that skips a bucket
// uses the same unfused fp32 operations as the distance, each monotone under rounding, so a
// skipped bucket provably holds no point whose running distance would change.
#include "common.h"
#include "fps_sort.h"

namespace prcnn {

constexpr int FB_CAP = 16384;          // point slots per cloud in the workspace (n <= FB_CAP)
constexpr int FB_BUCKET = 128;         // points per bucket: two per lane
constexpr int FB_NBUCKET = FB_CAP / FB_BUCKET;

typedef float fb_f4 __attribute__((ext_vector_type(4)));
typedef int fb_i2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct FbWs {
    fb_f4 *pts;      // (b, FB_CAP): x, y, z, tie-break key (int bits) in sorted order; padding slots zero
    int *td;         // (b, FB_CAP): running min distances (float bits); padding slots -1.0f
};
static size_t fb_bytes(int b) { return (size_t)b * FB_CAP * (sizeof(fb_f4) + sizeof(int)); }
static FbWs fb_carve(void *ws, int b)
{
    FbWs w;
    char *p = static_cast<char *>(ws);
    w.pts = reinterpret_cast<fb_f4 *>(p);
    w.td = reinterpret_cast<int *>(p + (size_t)b * FB_CAP * sizeof(fb_f4));
    return w;
}

// ---- 1. sort the cloud into the workspace (one 1024-thread workgroup per cloud, ~40 us) ----------
__global__ __launch_bounds__(1024) void fb_sort_kernel(int n, const float *__restrict__ dataset,
                                                        fb_f4 *__restrict__ pts_all, int *__restrict__ td_all)
{
    __shared__ FpsSortLds<16> s;
    __shared__ unsigned short order[FB_CAP];
    const int tid = threadIdx.x;
    const float *data = dataset + (size_t)blockIdx.x * n * 3;
    fps_grid_sort<16>(data, n, s, order);
    fb_f4 *pts = pts_all + (size_t)blockIdx.x * FB_CAP;
    int *td = td_all + (size_t)blockIdx.x * FB_CAP;
    for (int r = tid; r < FB_CAP; r += 1024) {
        const bool ok = r < n;
        int k = ok ? (int)order[r] : 0;
        k = k < n ? k : 0;                                     // (always true; keeps the read in bounds by construction)
        const int key = ((k & 511) << 8) | (k >> 9);
        fb_f4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) { v[0] = data[k * 3 + 0]; v[1] = data[k * 3 + 1]; v[2] = data[k * 3 + 2]; v[3] = i2f(key); }
        pts[r] = v;
        td[r] = ok ? f2i(1e38f) : f2i(-1.0f);
    }
}

__device__ __forceinline__ int fb_wave_min_i32(int v) { return ~wave_max_i32(~v); }

// ---- 2. the sampling chain: NW waves per cloud, bucket p * NW + wave on lane p of its wave -------
// One small workgroup per cloud, placed wherever the dispatcher finds room.  Measured alternatives
// (DESIGN.md 4.1): 2 waves per cloud (longer chain, twice the issue load on the SIMDs it touches:
// 10.9 k scenes/s in the pipelined bench against 12.1 k), 8 waves (12.0 k), and 16-wave workgroups
// that step through 2 / 4 / 8 clouds in lockstep and take a whole CU each (11.5 / 11.3 / 8.7 k).
template <int NW>
__global__ __launch_bounds__(NW * 64) void fb_sample_kernel(int n, int m, const float *__restrict__ dataset,
                                                          const fb_f4 *__restrict__ pts_all,
                                                          int *__restrict__ td_all, int *__restrict__ idxs)
{
    constexpr int BPW = FB_NBUCKET / NW;                       // buckets per wave
    static_assert(BPW >= 1 && BPW <= 64 && BPW * NW == FB_NBUCKET, "one bucket per lane");
    __shared__ unsigned long long slot[3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cloud = blockIdx.x;
    const float *cdata = dataset + (size_t)cloud * n * 3;      // winner coordinates (original indices)
    const fb_f4 *pts = pts_all + (size_t)cloud * FB_CAP;
    fb_i2 *td2 = reinterpret_cast<fb_i2 *>(td_all + (size_t)cloud * FB_CAP);
    int *out = idxs + (size_t)cloud * m;

    // first slot of bucket p of this wave, for this lane's pair of points (always < FB_CAP)
    auto slot_of = [&](int p) { return (((p * NW + wave) & (FB_NBUCKET - 1)) << 7) + 2 * lane; };

    // lane p < BPW holds bucket p of this wave: box, Tv = max running distance (bits; -1.0f =
    // empty), Nk = ~key of its holder; other lanes hold the identity of the (Tv, Nk) maximum
    float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY, bz0 = INFINITY, bz1 = -INFINITY;
    int Tv = (int)0x80000000, Nk = 0;
    for (int p = 0; p < BPW; ++p) {
        const int r0 = slot_of(p);
        const fb_f4 c0 = pts[r0], c1 = pts[r0 + 1];
        const fb_i2 t = td2[r0 >> 1];
        const bool ok0 = t[0] >= 0, ok1 = t[1] >= 0;           // padding slots carry -1.0f
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        int kmin = 0x7fffffff;
        if (ok0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], c0[d]); hi[d] = fmaxf(hi[d], c0[d]); }
            kmin = min(kmin, f2i(c0[3]));
        }
        if (ok1) {
#pragma unroll
            for (int d = 0; d < 3; ++d) { lo[d] = fminf(lo[d], c1[d]); hi[d] = fmaxf(hi[d], c1[d]); }
            kmin = min(kmin, f2i(c1[3]));
        }
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
                hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
            }
        const int k0 = __builtin_amdgcn_readfirstlane(fb_wave_min_i32(kmin));
        if (lane == p) {
            bx0 = lo[0]; bx1 = hi[0]; by0 = lo[1]; by1 = hi[1]; bz0 = lo[2]; bz1 = hi[2];
            Tv = k0 != 0x7fffffff ? f2i(1e38f) : f2i(-1.0f);
            Nk = ~k0;
        }
    }
    // maximum of (Tv signed : Nk unsigned) over the bucket lanes -> packed slot value of this wave:
    // the largest Tv first (one 32-bit DPP reduction), then the largest Nk among its holders
    // (mostly a single lane: one readlane)
    auto wave_best = [&](unsigned &hiw, unsigned &low) {
        const int tmax = wave_max_i32(Tv);                     // lanes >= BPW hold INT_MIN
        const unsigned long long holders = __ballot(Tv == tmax);
        if ((holders & (holders - 1)) == 0)
            low = (unsigned)__builtin_amdgcn_readlane(Nk, __builtin_ctzll(holders));
        else                                                    // unsigned max through the signed reduction
            low = (unsigned)wave_max_i32(Tv == tmax ? (int)((unsigned)Nk ^ 0x80000000u) : (int)0x80000000) ^ 0x80000000u;
        hiw = tmax < 0 ? 0u : (unsigned)tmax + 1u;             // fps_pack (fps.hip)
    };
    unsigned wb_hi, wb_lo;
    wave_best(wb_hi, wb_lo);
    if (tid < 3) slot[tid] = 0ull;
    if (tid == 0) out[0] = 0;
    int old = 0;                                               // wave-uniform
    __syncthreads();

    struct Bucket {              // this lane's pair of points of one bucket, as loaded
        fb_f4 c0, c1;
        fb_i2 t;
        int p, r0;
    };
    for (int j = 1; j < m; ++j) {
        const float x1 = cdata[old * 3 + 0], y1 = cdata[old * 3 + 1], z1 = cdata[old * 3 + 2];
        // lower bound of the distance from the sample to the box of bucket `lane`
        const float gx = fmaxf(fmaxf(bx0 - x1, x1 - bx1), 0.f);
        const float gy = fmaxf(fmaxf(by0 - y1, y1 - by1), 0.f);
        const float gz = fmaxf(fmaxf(bz0 - z1, z1 - bz1), 0.f);
        const float lb = gx * gx + gy * gy + gz * gz;
        unsigned long long dm = __ballot(f2i(lb) < Tv);
        if (BPW < 64) dm &= (1ull << (BPW & 63)) - 1ull;       // (lanes >= BPW hold INT_MIN and never vote)
        if (dm) {
            bool changed = false;
            auto fetch = [&](Bucket &k) {                      // next touched bucket (dm != 0)
                k.p = __builtin_ctzll(dm);
                dm &= dm - 1;
                k.r0 = slot_of(k.p);
                k.c0 = pts[k.r0]; k.c1 = pts[k.r0 + 1];
                k.t = td2[k.r0 >> 1];
            };
            auto work = [&](const Bucket &k) {
                // both points per packed instruction; each component is an IEEE fp32 operation, so the
                // result equals the scalar (dx*dx + dy*dy) + dz*dz bit for bit (as in fps.hip)
                const f32x2 px = {k.c0[0], k.c1[0]}, py = {k.c0[1], k.c1[1]}, pz = {k.c0[2], k.c1[2]};
                const f32x2 dx = px - x1, dy = py - y1, dz = pz - z1;
                const f32x2 d = dx * dx + dy * dy + dz * dz;
                const int t0 = k.t[0], t1 = k.t[1];
                const int n0 = min(f2i(d[0]), t0), n1 = min(f2i(d[1]), t1);
                if (n0 != t0 || n1 != t1) td2[k.r0 >> 1] = fb_i2{n0, n1};
                const int Tp = __builtin_amdgcn_readlane(Tv, k.p);
                if (__ballot((t0 == Tp && n0 != t0) || (t1 == Tp && n1 != t1))) {
                    // a holder of the bucket maximum was lowered: recompute (T, key)
                    const int v = max(n0, n1);
                    const int k0 = f2i(k.c0[3]), k1 = f2i(k.c1[3]);
                    const int kk = n0 > n1 ? k0 : (n1 > n0 ? k1 : min(k0, k1));
                    const int wm = wave_max_i32(v);
                    const unsigned long long holders = __ballot(v == wm);
                    int wk;
                    if ((holders & (holders - 1)) == 0)
                        wk = __builtin_amdgcn_readlane(kk, __builtin_ctzll(holders));
                    else
                        wk = fb_wave_min_i32(v == wm ? kk : 0x7fffffff);
                    Tv = lane == k.p ? wm : Tv;
                    Nk = lane == k.p ? ~wk : Nk;
                    changed = true;
                }
            };
            // two register sets in turn: the next touched bucket's loads fly while this one is worked on
            Bucket ka, kb;
            fetch(ka);
            for (;;) {
                const bool more_b = dm != 0;
                if (more_b) fetch(kb);
                work(ka);
                if (!more_b) break;
                const bool more_a = dm != 0;
                if (more_a) fetch(ka);
                work(kb);
                if (!more_a) break;
            }
            if (changed) wave_best(wb_hi, wb_lo);
        }
        const int cur = j % 3;
        if (lane == 0) atomicMax(&slot[cur], ((unsigned long long)wb_hi << 32) | wb_lo);
        __syncthreads();
        const unsigned long long win = slot[cur];
        if (tid == 0) slot[(j + 2) % 3] = 0ull;
        const unsigned k2 = ~(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)win) & 0x1ffffu;
        old = (int)((k2 >> 8) + ((k2 & 255u) << 9));
        old = old < n ? old : 0;                               // (always true; keeps the coordinate read in bounds)
        if (tid == 0) out[j] = old;
    }
}

}  // namespace prcnn

extern "C" size_t prcnn_fps_shared_temp_bytes(int b, int n)
{
    if (b <= 0 || n <= 0 || n > prcnn::FB_CAP) return 0;
    return prcnn::fb_bytes(b);
}

extern "C" int prcnn_farthest_point_sample_shared(int b, int n, int m, const float *inp, void *temp,
                                                  size_t temp_bytes, int *out, prcnn_stream_t stream)
{
    using namespace prcnn;
    if (m <= 0) return PRCNN_BAD_ARG;                 // tf_sampling.cpp:99 npoint > 0
    if (b < 0 || n <= 0) return PRCNN_BAD_SHAPE;      // tf_sampling.cpp:105
    if (n > FB_CAP) return PRCNN_UNSUPPORTED;
    if (b == 0) return PRCNN_OK;
    if (!inp || !out) return PRCNN_BAD_SHAPE;
    if (!temp || temp_bytes < fb_bytes(b)) return PRCNN_WORKSPACE;
    FbWs w = fb_carve(temp, b);
    if (((uintptr_t)temp & 15) != 0) return PRCNN_BAD_ARG;
    hipStream_t st = to_stream(stream);
    hipLaunchKernelGGL(fb_sort_kernel, dim3(b), dim3(1024), 0, st, n, inp, w.pts, w.td);
    static const int nw_env = getenv("PRCNN_FPS_SHARED_NW") ? atoi(getenv("PRCNN_FPS_SHARED_NW")) : 0;   // tuning aid
    if (nw_env == 8)
        hipLaunchKernelGGL((fb_sample_kernel<8>), dim3(b), dim3(512), 0, st, n, m, inp, w.pts, w.td, out);
    else if (nw_env == 2)
        hipLaunchKernelGGL((fb_sample_kernel<2>), dim3(b), dim3(128), 0, st, n, m, inp, w.pts, w.td, out);
    else
        hipLaunchKernelGGL((fb_sample_kernel<4>), dim3(b), dim3(256), 0, st, n, m, inp, w.pts, w.td, out);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}
