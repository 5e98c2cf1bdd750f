// Detection records for the scene-sharded harness: one fixed-size float32
// record per scene, [scene_id, n_keep, MAX_DET x (x, y, z, l, h, w, ry, score)], zero past
// n_keep -- what each rank contributes to the single all-gather per step.  One launch instead
// of a dozen elementwise / gather / cat kernels at the tail of every step.
#include "common.h"

namespace prcnn {

__global__ void pack_detections_kernel(int scenes, int nprop, int max_det, const float *__restrict__ scene_ids,
                                       const double *__restrict__ boxes3d, const double *__restrict__ scores,
                                       const long long *__restrict__ keep, const int *__restrict__ n_keep,
                                       float *__restrict__ rec)
{
    const int rl = 2 + max_det * 8;
    const long long total = (long long)scenes * rl;
    for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(e / rl);
        const int f = (int)(e - (long long)s * rl);
        const int nk = min(max(n_keep[s], 0), max_det);
        float v;
        if (f == 0) {
            v = scene_ids[s];
        } else if (f == 1) {
            v = (float)n_keep[s];
        } else {
            const int d = (f - 2) >> 3, c = (f - 2) & 7;
            v = 0.f;
            if (d < nk) {
                const long long k = keep[(size_t)s * max_det + d];
                if (k >= 0 && k < nprop)
                    v = (float)(c < 7 ? boxes3d[((size_t)s * nprop + k) * 7 + c] : scores[(size_t)s * nprop + k]);
            }
        }
        rec[e] = v;
    }
}

}  // namespace prcnn

extern "C" int prcnn_pack_detections(int scenes, int nprop, int max_det, const float *scene_ids,
                                     const double *boxes3d, const double *scores, const int64_t *keep,
                                     const int *n_keep, float *rec, prcnn_stream_t stream)
{
    if (scenes < 0 || nprop < 0 || max_det < 0) return PRCNN_BAD_SHAPE;
    if (scenes == 0) return PRCNN_OK;
    if (!scene_ids || !n_keep || !rec) return PRCNN_BAD_SHAPE;
    if (max_det > 0 && (!keep || (nprop > 0 && (!boxes3d || !scores)))) return PRCNN_BAD_SHAPE;
    const long long total = (long long)scenes * (2 + max_det * 8);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(prcnn::pack_detections_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream),
                       scenes, nprop, max_det, scene_ids, boxes3d, scores, (const long long *)keep, n_keep, rec);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}
