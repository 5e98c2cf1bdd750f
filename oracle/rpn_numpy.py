"""NumPy restatement of the RPN proposal post-processing (TEST INFRASTRUCTURE ONLY).

Follows, line by line, TensorFlow graph code of the reference (no native kernel there):
  tf_decode                   utils/box_encoder.py:109-179
  box_center + fg_points_xyz  models/rpn.py:108-109
  get_box3d_corners_helper    models/model_util.py:55-82
  reduce_proposals            models/rpn.py:209-241
tf.image.non_max_suppression is TensorFlow's NonMaxSuppression kernel (third-party,
un-vendored: TF 1.x core/kernels/non_max_suppression_op.cc), restated from its published
algorithm.  Parity unpinned: TensorFlow is not installed and the reference holds no fixture
for these functions; the restatement is anchored on the reference's call sites only.
"""
import numpy as np

TYPE_MEAN_SIZE = np.array([[3.88311640418, 1.52563191462, 1.62856739989],      # box_encoder.py:6-8 (l,h,w)
                           [0.84422524, 1.76255119, 0.66068622],
                           [1.76282397, 1.73698127, 0.59706367]], np.float32)


def _softmax_max(x):
    x = x.astype(np.float32)
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m, dtype=np.float32)
    return (np.float32(1.0) / e.sum(-1, dtype=np.float32)).astype(np.float32)


def split_head(output, ncb=12, nhb=12, nsc=3):
    """models/rpn.py:76-104."""
    o = 0
    ep = {}
    ep['center_x_scores'] = output[..., o:o + ncb]; o += ncb
    ep['center_x_residuals_normalized'] = output[..., o:o + ncb]; o += ncb
    ep['center_z_scores'] = output[..., o:o + ncb]; o += ncb
    ep['center_z_residuals_normalized'] = output[..., o:o + ncb]; o += ncb
    ep['center_y_residuals'] = output[..., o:o + 1]; o += 1
    ep['heading_scores'] = output[..., o:o + nhb]; o += nhb
    ep['heading_residuals_normalized'] = output[..., o:o + nhb]; o += nhb
    ep['size_scores'] = output[..., o:o + nsc]; o += nsc
    ep['size_residuals_normalized'] = output[..., o:o + nsc * 3].reshape(output.shape[:-1] + (nsc, 3))
    return ep


def tf_decode(ep, center_range=3.0, ncb=12, heading_range=np.pi, nhb=12, mean_size=TYPE_MEAN_SIZE):
    f = np.float32
    take = lambda a, i: np.take_along_axis(a, i[..., None], -1)[..., 0]
    bin_x = ep['center_x_scores'].argmax(-1)
    bin_z = ep['center_z_scores'].argmax(-1)
    bin_size = f(center_range * 2 / ncb)
    rng = f(center_range)
    cx = (bin_x.astype(f) * bin_size + bin_size / f(2) - rng) + take(ep['center_x_residuals_normalized'], bin_x) * bin_size
    cy = f(0) + ep['center_y_residuals'][..., 0]
    cz = (bin_z.astype(f) * bin_size + bin_size / f(2) - rng) + take(ep['center_z_residuals_normalized'], bin_z) * bin_size
    center = np.stack([cx, cy, cz], -1).astype(f)
    hcls = ep['heading_scores'].argmax(-1)
    apc = f(2 * heading_range / float(nhb))
    angle = (hcls.astype(f) * apc + take(ep['heading_residuals_normalized'], hcls) * apc) - f(heading_range)
    scls = ep['size_scores'].argmax(-1)
    mean = mean_size.astype(f)[scls]
    res = np.take_along_axis(ep['size_residuals_normalized'], scls[..., None, None], -2)[..., 0, :]
    size = (mean + res * mean).astype(f)
    return center, angle.astype(f), size


def get_box3d_corners(centers, headings, sizes):
    f = np.float32
    l, h, w = sizes[..., 0], sizes[..., 1], sizes[..., 2]
    z = np.zeros_like(h)
    xc = np.stack([l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2], -1)
    yc = np.stack([z, z, z, z, -h, -h, -h, -h], -1)
    zc = np.stack([w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2], -1)
    c, s = np.cos(headings, dtype=f), np.sin(headings, dtype=f)
    X = (c[..., None] * xc + s[..., None] * zc) + centers[..., 0:1]
    Y = yc + centers[..., 1:2]
    Z = (-s[..., None] * xc + c[..., None] * zc) + centers[..., 2:3]
    return np.stack([X, Y, Z], -1).astype(f)            # (...,8,3)


def tf_nms(boxes, scores, max_output_size, iou_threshold=0.5):
    """TensorFlow NonMaxSuppression; equal scores: lower index first (TF: unspecified)."""
    f = np.float32
    boxes = boxes.astype(f)
    order = np.argsort(-scores.astype(np.float64), kind='stable')
    ymin = np.minimum(boxes[:, 0], boxes[:, 2]); ymax = np.maximum(boxes[:, 0], boxes[:, 2])
    xmin = np.minimum(boxes[:, 1], boxes[:, 3]); xmax = np.maximum(boxes[:, 1], boxes[:, 3])
    area = ((ymax - ymin) * (xmax - xmin)).astype(f)
    sel = []
    for i in order:
        if len(sel) >= max_output_size:
            break
        ok = True
        for j in reversed(sel):
            if area[i] <= 0 or area[j] <= 0:
                continue
            iy = max(f(min(ymax[i], ymax[j]) - max(ymin[i], ymin[j])), f(0))
            ix = max(f(min(xmax[i], xmax[j]) - max(xmin[i], xmin[j])), f(0))
            inter = f(iy * ix)
            if f(inter / f(f(area[i] + area[j]) - inter)) > f(iou_threshold):
                ok = False
                break
        if ok:
            sel.append(int(i))
    return np.array(sel, np.int64)


def decode_and_reduce(output, fg_points_xyz, fg_logits, max_out=300, iou=0.5, **kw):
    """rpn.py:108-113 + reduce_proposals :209-241 -> dict like the reference's end_points."""
    ep = split_head(output, kw.get('ncb', 12), kw.get('nhb', 12), kw.get('nsc', 3))
    center, angle, size = tf_decode(ep, **{k: v for k, v in kw.items() if k in ('center_range', 'ncb', 'heading_range', 'nhb')})
    center = (center + fg_points_xyz).astype(np.float32)
    corners = get_box3d_corners(center, angle, size)
    conf = (_softmax_max(fg_logits) * _softmax_max(ep['center_x_scores']) * _softmax_max(ep['center_z_scores'])
            * _softmax_max(ep['heading_scores']) * _softmax_max(ep['size_scores'])).astype(np.float32)
    mn, mx = corners.min(-2), corners.max(-2)
    bev = np.stack([mn[..., 0], mn[..., 2], mx[..., 0], mx[..., 2]], -1).astype(np.float32)
    B, M = conf.shape
    nms = np.full((B, M), -1, np.int32)
    for b in range(B):
        k = tf_nms(bev[b], conf[b], max_out, iou)
        nms[b, :len(k)] = k
    return {'center': center, 'angle': angle, 'size': size, 'proposal_boxes': corners,
            'proposal_scores': conf, 'boxes_bev': bev, 'nms_indices': nms}
