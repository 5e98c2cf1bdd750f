// RPN proposal post-processing around the rotated NMS.
//
// Reference (all TensorFlow graph ops, no native kernel):
//   * BoxEncoder.tf_decode            utils/box_encoder.py:109-179   per-point bin argmax ->
//     centre / heading / size;  models/rpn.py:108-109 adds the point itself to the centre;
//   * get_box3d_corners_helper        models/model_util.py:55-82     8 corners, rotation about y;
//   * RPN.reduce_proposals            models/rpn.py:209-241          confidence = product of the
//     max-softmax of the five heads, BEV box = min/max of the corners' (x, z).
// One lane per proposal does all of it in one pass over the head output row:
//   row = [cx scores (NCB) | cx residuals (NCB) | cz scores | cz residuals | cy residual (1) |
//          heading scores (NHB) | heading residuals (NHB) | size scores (NSC) | size residuals (NSC*3)]
//   (models/rpn.py:76-104).  argmax = first maximum (tf.argmax); max softmax = 1 / sum exp(x - max).
#include "common.h"

namespace prcnn {

struct RpnHeadArgs {
    int ncb, nhb, nsc;
    float center_range, heading_range;
    float mean_size[8 * 3];                // (l, h, w) per size cluster
    const float *head, *fg_xyz, *fg_logits;
    float *center, *angle, *size, *corners, *confidence, *bev;
};
constexpr int RPN_MAX_CLUSTERS = 8;

__device__ __forceinline__ void argmax_softmax(const float *x, int n, int &arg, float &pmax)
{
    int a = 0;
    float m = x[0];
    for (int i = 1; i < n; ++i)
        if (x[i] > m) { m = x[i]; a = i; }          // first maximum
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += expf(x[i] - m);
    arg = a;
    pmax = 1.0f / s;
}

__global__ void rpn_decode_kernel(long long total, int row_len, const RpnHeadArgs a)
{
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < total;
         p += (long long)gridDim.x * blockDim.x) {
        const float *row = a.head + p * row_len;
        const float *cxs = row, *cxr = row + a.ncb, *czs = row + 2 * a.ncb, *czr = row + 3 * a.ncb;
        const float cyr = row[4 * a.ncb];
        const float *hs = row + 4 * a.ncb + 1, *hr = hs + a.nhb;
        const float *ss = hr + a.nhb, *sr = ss + a.nsc;
        int bx, bz, bh, bs;
        float px, pz, ph, ps;
        argmax_softmax(cxs, a.ncb, bx, px);
        argmax_softmax(czs, a.ncb, bz, pz);
        argmax_softmax(hs, a.nhb, bh, ph);
        argmax_softmax(ss, a.nsc, bs, ps);
        // centre (box_encoder.py:134-157, rpn.py:109)
        const float bin = a.center_range * 2.f / (float)a.ncb;
        const float cx = ((float)bx * bin + bin / 2.f - a.center_range) + cxr[bx] * bin + a.fg_xyz[p * 3 + 0];
        const float cy = (0.f + cyr) + a.fg_xyz[p * 3 + 1];
        const float cz = ((float)bz * bin + bin / 2.f - a.center_range) + czr[bz] * bin + a.fg_xyz[p * 3 + 2];
        // heading (box_encoder.py:158-165)
        const float apc = 2.f * a.heading_range / (float)a.nhb;
        const float ang = ((float)bh * apc + hr[bh] * apc) - a.heading_range;
        // size (box_encoder.py:166-177): mean + residual * mean
        float sz[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float mean = a.mean_size[bs * 3 + d];
            sz[d] = mean + sr[bs * 3 + d] * mean;
        }
        if (a.center) { a.center[p * 3] = cx; a.center[p * 3 + 1] = cy; a.center[p * 3 + 2] = cz; }
        if (a.angle) a.angle[p] = ang;
        if (a.size) { a.size[p * 3] = sz[0]; a.size[p * 3 + 1] = sz[1]; a.size[p * 3 + 2] = sz[2]; }
        // corners (model_util.py:55-82): R = [[c,0,s],[0,1,0],[-s,0,c]]
        const float l = sz[0], h = sz[1], w = sz[2];
        const float c = cosf(ang), s = sinf(ang);
        const float xs[8] = {l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2};
        const float ys[8] = {0.f, 0.f, 0.f, 0.f, -h, -h, -h, -h};
        const float zs[8] = {w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2};
        float xmin = INFINITY, xmax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float X = (c * xs[k] + 0.f * ys[k] + s * zs[k]) + cx;
            const float Y = (0.f * xs[k] + 1.f * ys[k] + 0.f * zs[k]) + cy;
            const float Z = (-s * xs[k] + 0.f * ys[k] + c * zs[k]) + cz;
            if (a.corners) {
                a.corners[(p * 8 + k) * 3 + 0] = X;
                a.corners[(p * 8 + k) * 3 + 1] = Y;
                a.corners[(p * 8 + k) * 3 + 2] = Z;
            }
            xmin = fminf(xmin, X); xmax = fmaxf(xmax, X);
            zmin = fminf(zmin, Z); zmax = fmaxf(zmax, Z);
        }
        // rpn.py:224-227: [min x, min z, max x, max z]
        if (a.bev) { a.bev[p * 4] = xmin; a.bev[p * 4 + 1] = zmin; a.bev[p * 4 + 2] = xmax; a.bev[p * 4 + 3] = zmax; }
        // rpn.py:213-220
        if (a.confidence) {
            float pseg = 1.f;
            if (a.fg_logits) {
                int t;
                argmax_softmax(a.fg_logits + p * 2, 2, t, pseg);
            }
            a.confidence[p] = pseg * px * pz * ph * ps;
        }
    }
}

}  // namespace prcnn

extern "C" int prcnn_rpn_decode_proposals(int b, int n, int num_center_bin, int num_heading_bin,
                                          int num_size_cluster, float center_search_range,
                                          float heading_search_range, const float *host_mean_sizes,
                                          const float *head_output, const float *fg_points_xyz,
                                          const float *fg_logits, float *center, float *angle, float *size,
                                          float *corners, float *confidence, float *bev_boxes,
                                          prcnn_stream_t stream)
{
    using namespace prcnn;
    if (b < 0 || n < 0) return PRCNN_BAD_SHAPE;
    if (num_center_bin <= 0 || num_heading_bin <= 0 || num_size_cluster <= 0) return PRCNN_BAD_ARG;
    if (num_size_cluster > RPN_MAX_CLUSTERS) return PRCNN_UNSUPPORTED;
    const long long total = (long long)b * n;
    if (total == 0) return PRCNN_OK;
    if (!host_mean_sizes || !head_output || !fg_points_xyz) return PRCNN_BAD_SHAPE;
    RpnHeadArgs a = {};
    a.ncb = num_center_bin; a.nhb = num_heading_bin; a.nsc = num_size_cluster;
    a.center_range = center_search_range; a.heading_range = heading_search_range;
    for (int i = 0; i < num_size_cluster * 3; ++i) a.mean_size[i] = host_mean_sizes[i];
    a.head = head_output; a.fg_xyz = fg_points_xyz; a.fg_logits = fg_logits;
    a.center = center; a.angle = angle; a.size = size; a.corners = corners; a.confidence = confidence;
    a.bev = bev_boxes;
    const int row_len = 4 * num_center_bin + 1 + 2 * num_heading_bin + 4 * num_size_cluster;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rpn_decode_kernel, dim3((unsigned)blocks), dim3(256), 0, to_stream(stream), total,
                       row_len, a);
    PRCNN_LAUNCH_CHECK();
    return PRCNN_OK;
}
