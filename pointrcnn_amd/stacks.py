"""The SA/FP stacks of the reference's models, expressed with pointnet_util.

Only the set-abstraction / feature-propagation backbones are here (the hot path); heads,
losses and box decoding are out of scope.  Hyper-parameters are the
reference's literals:
  rpn_sa_fp_stack        models/rpn.py:159-192           (RPN backbone, 4 SA-MSG + 4 FP)
  frustum_cls_stack      models/frustum_pointnets_v2.py:131-142
  frustum_seg_stack      models/frustum_pointnets_v2.py:212-234
  frustum_box_stack      models/frustum_pointnets_v2.py:272-283
"""
import torch

from .pointnet_util import pointnet_fp_module, pointnet_sa_module, pointnet_sa_module_msg

# (npoint, radii, nsamples, mlps) per SA-MSG level, models/rpn.py:166-181
RPN_SA = [
    (4096, [0.1, 0.5], [16, 32], [[16, 16, 32], [32, 32, 64]]),
    (1024, [0.5, 1.0], [16, 32], [[64, 64, 128], [64, 96, 128]]),
    (256, [1.0, 2.0], [16, 32], [[128, 196, 256], [128, 196, 256]]),
    (64, [2.0, 4.0], [16, 32], [[256, 256, 512], [256, 384, 512]]),
]
# mlp per FP level, models/rpn.py:184-192
RPN_FP = [[512, 512], [512, 512], [256, 256], [128, 128]]


def sample_centroid_chain(l0_xyz, npoints, stream=None, shared=None):
    """FPS + gather for every level: l0 -> l1 -> ... depends on coordinates only, so the
    whole (strictly serial, latency-bound) chain can run on a side stream while the main
    stream does the feature work of the levels already sampled.  Returns the list of
    new_xyz tensors and one event per level recorded on `stream`.
    shared=True: long chains over 4096 < n <= 16384 points (level 1) run on the small-footprint
    sampler, which leaves its CU to the kernels in flight beside it (same indices; slower alone)."""
    from . import ops
    xs, evs = [], []
    cur = torch.cuda.current_stream()
    side = stream if stream is not None else cur
    if stream is not None:
        side.wait_stream(cur)
    with torch.cuda.stream(side):
        x = l0_xyz
        for m in npoints:
            sh = (shared and 4096 < x.shape[1] <= 16384 and m >= 256) if shared is not None else None
            x = ops.gather_point(x, ops.farthest_point_sample(m, x, shared=sh))
            xs.append(x)
            if stream is not None:
                ev = torch.cuda.Event()
                ev.record(side)
                evs.append(ev)
                x.record_stream(cur)
    return xs, evs


# SHARE_ORDER: the level-1 ball query (points = the cloud, centroids = level-1 new_xyz) and the last FP
# level's three_nn (unknown = the cloud, known = level-1 new_xyz) search the same two point sets; when
# both take the culled path, the grid order is computed once (ops.grid_order) and handed to both.
SHARE_ORDER = True


def shared_order(l0_xyz, l1_xyz):
    from . import ops
    if not SHARE_ORDER:
        return None
    bq, nn = ops.culled_pair(l0_xyz.shape[1], l1_xyz.shape[1], RPN_SA[0][2])
    return ops.grid_order(l0_xyz, l1_xyz) if bq and nn else None


def rpn_geometry(l0_xyz, centroids):
    """Everything in the RPN stack that depends on coordinates only, given the four levels'
    centroids: the ball-query answers of every SA scale and, per FP level, three_nn + the
    inverse-distance weights.  Like the FPS chain it can run ahead of (and beside) the feature
    phase.  Returns {"balls": [[(idx, pts_cnt)] per scale] per SA level,
                     "nn": [(idx, weight)] per FP level in call order (fa_layer2..5)}."""
    from . import ops
    xyzs = [l0_xyz] + list(centroids)
    order01 = shared_order(xyzs[0], xyzs[1])
    balls = [ops.query_ball_point_msg(radii, nsamples, xyzs[li], xyzs[li + 1], order=order01 if li == 0 else None)
             for li, (_, radii, nsamples, _) in enumerate(RPN_SA)]
    nn = []
    for lo in (3, 2, 1, 0):
        dist, idx = ops.three_nn(xyzs[lo], xyzs[lo + 1], order=order01 if lo == 0 else None)
        nn.append((idx, ops.fp_weights(dist)))
    return {"balls": balls, "nn": nn}


def geometry_tensors(geo):
    """Flat list of the tensors of an rpn_geometry() result (fixed order)."""
    out = []
    for level in geo["balls"]:
        for idx, cnt in level:
            out += [idx, cnt]
    for idx, w in geo["nn"]:
        out += [idx, w]
    return out


def geometry_from_tensors(like, tensors):
    """Inverse of geometry_tensors: the structure of `like` over the flat list `tensors`."""
    it = iter(tensors)
    return {"balls": [[(next(it), next(it)) for _ in level] for level in like["balls"]],
            "nn": [(next(it), next(it)) for _ in like["nn"]]}


def flat_copy_set(tensors):
    """Copies of `tensors` laid out in ONE flat byte buffer (256-byte aligned pieces): returns
    (flat, views).  Two such sets of equal layout are copied into one another with one copy."""
    offs, o = [], 0
    for t in tensors:
        offs.append(o)
        o += (t.numel() * t.element_size() + 255) // 256 * 256
    flat = torch.empty(max(o, 256), dtype=torch.uint8, device=tensors[0].device)
    views = []
    for off, t in zip(offs, tensors):
        v = flat[off:off + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
        v.copy_(t)
        views.append(v)
    return flat, views


def geometry_clone(geo):
    return {"balls": [[(idx.clone(), cnt.clone()) for idx, cnt in level] for level in geo["balls"]],
            "nn": [(idx.clone(), w.clone()) for idx, w in geo["nn"]]}


def geometry_slice(geo, lo, hi):
    """The clouds [lo, hi) of a batched rpn_geometry() result (views)."""
    return {"balls": [[(idx[lo:hi], cnt[lo:hi]) for idx, cnt in level] for level in geo["balls"]],
            "nn": [(idx[lo:hi], w[lo:hi]) for idx, w in geo["nn"]]}


def rpn_sa_fp_stack(point_cloud, params=None, precision=None, is_training=False, bn_decay=None,
                    geometry_stream=None, centroids=None, geometry=None):
    """point_cloud (B,N,4) -> per-point features (B,N,128); also returns the level xyz/feature lists.
    With `geometry_stream` the FPS chain of all four levels runs on that stream, overlapped
    with the grouping / MLP work of the main stream (same results)."""
    l0_xyz = point_cloud[:, :, 0:3].contiguous()
    l0_points = point_cloud[:, :, 3:].contiguous() if point_cloud.shape[2] > 3 else None
    xyzs, feats = [l0_xyz], [l0_points]
    pre, evs = (None, None)
    if centroids is not None:
        # the four levels' new_xyz were sampled beforehand (sample_centroid_chain on the same
        # cloud, e.g. while the previous batch was in its feature phase): same results
        pre, evs = list(centroids), None
    elif geometry_stream is not None:
        pre, evs = sample_centroid_chain(l0_xyz, [lv[0] for lv in RPN_SA], geometry_stream)
    order01 = None
    for li, (npoint, radii, nsamples, mlps) in enumerate(RPN_SA):
        if evs is not None:
            torch.cuda.current_stream().wait_event(evs[li])
        new_xyz = None if pre is None else pre[li]
        if li == 0 and geometry is None and SHARE_ORDER:
            if new_xyz is None:
                from . import ops
                new_xyz = ops.gather_point(l0_xyz, ops.farthest_point_sample(npoint, l0_xyz))
            order01 = shared_order(l0_xyz, new_xyz)
        x, f = pointnet_sa_module_msg(xyzs[-1], feats[-1], npoint, radii, nsamples, mlps, is_training,
                                      bn_decay, scope='layer%d' % (li + 1), bn=True, params=params,
                                      precision=precision, new_xyz=new_xyz,
                                      balls=None if geometry is None else geometry["balls"][li],
                                      order=order01 if li == 0 else None)
        xyzs.append(x)
        feats.append(f)
    nn = [None] * 4 if geometry is None else geometry["nn"]
    l3 = pointnet_fp_module(xyzs[3], xyzs[4], feats[3], feats[4], RPN_FP[0], is_training, bn_decay,
                            scope='fa_layer2', bn=True, params=params, precision=precision, nn=nn[0])
    l2 = pointnet_fp_module(xyzs[2], xyzs[3], feats[2], l3, RPN_FP[1], is_training, bn_decay,
                            scope='fa_layer3', bn=True, params=params, precision=precision, nn=nn[1])
    l1 = pointnet_fp_module(xyzs[1], xyzs[2], feats[1], l2, RPN_FP[2], is_training, bn_decay,
                            scope='fa_layer4', bn=True, params=params, precision=precision, nn=nn[2])
    # concat([l0_xyz, l0_points]) (models/rpn.py:191) is the input cloud itself
    p1 = point_cloud.contiguous()
    l0 = pointnet_fp_module(l0_xyz, xyzs[1], p1, l1, RPN_FP[3], is_training, bn_decay,
                            scope='fa_layer5', bn=True, params=params, precision=precision, nn=nn[3],
                            order=order01)
    return l0, {"xyz": xyzs, "sa_feats": feats, "fp_feats": [l3, l2, l1, l0]}


def _frustum_msg_trunk(l0_xyz, l0_points, prefix, params, precision):
    l1_xyz, l1_points = pointnet_sa_module_msg(
        l0_xyz, l0_points, 128, [0.2, 0.4, 0.8], [32, 64, 128],
        [[32, 32, 64], [64, 64, 128], [64, 96, 128]], False, None, scope=prefix + 'layer1',
        params=params, precision=precision)
    l2_xyz, l2_points = pointnet_sa_module_msg(
        l1_xyz, l1_points, 32, [0.4, 0.8, 1.6], [64, 64, 128],
        [[64, 64, 128], [128, 128, 256], [128, 128, 256]], False, None, scope=prefix + 'layer2',
        params=params, precision=precision)
    return l1_xyz, l1_points, l2_xyz, l2_points


def frustum_cls_stack(point_cloud, params=None, precision=None):
    """models/frustum_pointnets_v2.py:131-142 -> global feature (B,512)."""
    l0_xyz = point_cloud[:, :, 0:3].contiguous()
    l0_points = point_cloud[:, :, 3:].contiguous()
    _, _, l2_xyz, l2_points = _frustum_msg_trunk(l0_xyz, l0_points, 'cls_', params, precision)
    _, l3_points, _ = pointnet_sa_module(l2_xyz, l2_points, npoint=None, radius=None, nsample=None,
                                         mlp=[128, 256, 512], mlp2=None, group_all=True,
                                         is_training=False, bn_decay=None, scope='cls_layer3',
                                         params=params, precision=precision)
    return l3_points.reshape(point_cloud.shape[0], -1)


def frustum_seg_stack(point_cloud, one_hot_vec, params=None, precision=None):
    """models/frustum_pointnets_v2.py:212-234 -> per-point features (B,N,128)."""
    l0_xyz = point_cloud[:, :, 0:3].contiguous()
    l0_points = point_cloud[:, :, 3:].contiguous()
    l1_xyz, l1_points, l2_xyz, l2_points = _frustum_msg_trunk(l0_xyz, l0_points, '', params, precision)
    l3_xyz, l3_points, _ = pointnet_sa_module(l2_xyz, l2_points, npoint=None, radius=None, nsample=None,
                                              mlp=[128, 256, 1024], mlp2=None, group_all=True,
                                              is_training=False, bn_decay=None, scope='layer3',
                                              params=params, precision=precision)
    l3_points = torch.cat([l3_points, one_hot_vec.unsqueeze(1)], dim=2)          # :226
    l2_points = pointnet_fp_module(l2_xyz, l3_xyz, l2_points, l3_points, [128, 128], False, None,
                                   scope='fa_layer1', params=params, precision=precision)
    l1_points = pointnet_fp_module(l1_xyz, l2_xyz, l1_points, l2_points, [128, 128], False, None,
                                   scope='fa_layer2', params=params, precision=precision)
    l0_points = pointnet_fp_module(l0_xyz, l1_xyz, torch.cat([l0_xyz, l0_points], dim=-1), l1_points,
                                   [128, 128], False, None, scope='fa_layer3', params=params,
                                   precision=precision)
    return l0_points


def frustum_box_stack(object_point_cloud, params=None, precision=None):
    """models/frustum_pointnets_v2.py:268-283 (SSG; first layer has points=None when the cloud
    is xyz only) -> global feature (B,512)."""
    l0_xyz = object_point_cloud[:, :, 0:3].contiguous()
    l0_points = (object_point_cloud[:, :, 3:].contiguous()
                 if object_point_cloud.shape[2] > 3 else None)
    l1_xyz, l1_points, _ = pointnet_sa_module(l0_xyz, l0_points, npoint=128, radius=0.2, nsample=64,
                                              mlp=[64, 64, 128], mlp2=None, group_all=False,
                                              is_training=False, bn_decay=None, scope='ssg-layer1',
                                              params=params, precision=precision)
    l2_xyz, l2_points, _ = pointnet_sa_module(l1_xyz, l1_points, npoint=32, radius=0.4, nsample=64,
                                              mlp=[128, 128, 256], mlp2=None, group_all=False,
                                              is_training=False, bn_decay=None, scope='ssg-layer2',
                                              params=params, precision=precision)
    _, l3_points, _ = pointnet_sa_module(l2_xyz, l2_points, npoint=None, radius=None, nsample=None,
                                         mlp=[256, 256, 512], mlp2=None, group_all=True,
                                         is_training=False, bn_decay=None, scope='ssg-layer3',
                                         params=params, precision=precision)
    return l3_points.reshape(object_point_cloud.shape[0], -1)
