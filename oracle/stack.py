"""CPU composition of the RPN SA+FP stack from the oracle ops (TEST INFRASTRUCTURE ONLY).

Follows models/rpn.py:159-192 and models/pointnet_util.py:155-228 of the reference op by
op (no fusion), with the inference-time BN folded like the product does.  Used by
tests/ (end-to-end stack parity) and by bench.py's cpu_baseline leg (the "reference CPU
path" of BASELINE.json: C++ CPU ops + transcribed FPS), never by the product.
`gemm='blas'` swaps the float64 triple loop for a NumPy fp32 matmul so that the timed
baseline is not handicapped by a naive GEMM (the reference would use Eigen/cuDNN there).
`gemm='f64blas'` is the float64 triple loop of oracle_mlp_layer evaluated by a float64 BLAS
matmul (products and sums in double, scale/shift/ReLU in double, one rounding to float32 per
layer): the same value to ~1e-15 relative, fast enough for the full-size stack tests.

Also here: the three Frustum PointNet v2 SA/FP stacks (models/frustum_pointnets_v2.py:131-142,
212-234, 272-283) with SSG / group_all set abstraction (models/pointnet_util.py:21-83,86-153).
"""
import numpy as np

from . import oracle as O

RPN_SA = [
    (4096, [0.1, 0.5], [16, 32], [[16, 16, 32], [32, 32, 64]]),
    (1024, [0.5, 1.0], [16, 32], [[64, 64, 128], [64, 96, 128]]),
    (256, [1.0, 2.0], [16, 32], [[128, 196, 256], [128, 196, 256]]),
    (64, [2.0, 4.0], [16, 32], [[256, 256, 512], [256, 384, 512]]),
]
RPN_FP = [[512, 512], [512, 512], [256, 256], [128, 128]]


def _layer(x, p, gemm):
    W, s, t = p
    if gemm == 'blas':
        y = x.reshape(-1, x.shape[-1]) @ W
        y = np.maximum(y * s + t, 0).astype(np.float32)
        return y.reshape(x.shape[:-1] + (W.shape[1],))
    if gemm == 'f64blas':
        y = x.reshape(-1, x.shape[-1]).astype(np.float64) @ W.astype(np.float64)
        y = np.maximum(y * s.astype(np.float64) + t.astype(np.float64), 0).astype(np.float32)
        return y.reshape(x.shape[:-1] + (W.shape[1],))
    return O.mlp_layer(x, W, s, t, True)


def sa_msg(xyz, points, npoint, radii, nsamples, mlps, scope, params, gemm='f64', timers=None):
    import time
    t0 = time.perf_counter()
    new_xyz = O.gather_point(xyz, O.farthest_point_sample(npoint, xyz))
    if timers is not None:
        timers['fps'] = timers.get('fps', 0) + time.perf_counter() - t0
    outs = []
    for i, (r, ns) in enumerate(zip(radii, nsamples)):
        t0 = time.perf_counter()
        idx, _ = O.query_ball_point(r, ns, xyz, new_xyz)
        t1 = time.perf_counter()
        g = O.group_point(xyz, idx) - new_xyz[:, :, None, :]
        if points is not None:
            g = np.concatenate([O.group_point(points, idx), g], -1)
        t2 = time.perf_counter()
        for j in range(len(mlps[i])):
            g = _layer(g, params['%s/conv%d_%d' % (scope, i, j)], gemm)
        outs.append(g.max(axis=2))
        t3 = time.perf_counter()
        if timers is not None:
            timers['ball_query'] = timers.get('ball_query', 0) + t1 - t0
            timers['group'] = timers.get('group', 0) + t2 - t1
            timers['mlp'] = timers.get('mlp', 0) + t3 - t2
    return new_xyz, np.concatenate(outs, -1)


def fp(xyz1, xyz2, points1, points2, mlp, scope, params, gemm='f64', timers=None):
    import time
    t0 = time.perf_counter()
    d, i3 = O.three_nn(xyz1, xyz2)
    t1 = time.perf_counter()
    interp = O.three_interpolate(points2, i3, O.fp_weights(d))
    x = np.concatenate([interp, points1], -1) if points1 is not None else interp
    t2 = time.perf_counter()
    for j in range(len(mlp)):
        x = _layer(x, params['%s/conv_%d' % (scope, j)], gemm)
    t3 = time.perf_counter()
    if timers is not None:
        timers['three_nn'] = timers.get('three_nn', 0) + t1 - t0
        timers['interpolate'] = timers.get('interpolate', 0) + t2 - t1
        timers['mlp'] = timers.get('mlp', 0) + t3 - t2
    return x


def rpn_sa_fp_stack(pc, params, gemm='f64', timers=None, sa=RPN_SA, fpm=RPN_FP):
    """pc (B,N,4) float32, params: scope name -> (W, scale, shift) NumPy arrays."""
    xyz0 = np.ascontiguousarray(pc[:, :, :3])
    p0 = np.ascontiguousarray(pc[:, :, 3:]) if pc.shape[2] > 3 else None
    xyzs, feats = [xyz0], [p0]
    for li, (npoint, radii, nsamples, mlps) in enumerate(sa):
        x, f = sa_msg(xyzs[-1], feats[-1], npoint, radii, nsamples, mlps, 'layer%d' % (li + 1), params,
                      gemm, timers)
        xyzs.append(x); feats.append(f)
    l3 = fp(xyzs[3], xyzs[4], feats[3], feats[4], fpm[0], 'fa_layer2', params, gemm, timers)
    l2 = fp(xyzs[2], xyzs[3], feats[2], l3, fpm[1], 'fa_layer3', params, gemm, timers)
    l1 = fp(xyzs[1], xyzs[2], feats[1], l2, fpm[2], 'fa_layer4', params, gemm, timers)
    p1 = np.concatenate([xyz0, p0], -1) if p0 is not None else xyz0
    l0 = fp(xyz0, xyzs[1], p1, l1, fpm[3], 'fa_layer5', params, gemm, timers)
    return l0, {"xyz": xyzs, "sa_feats": feats, "fp_feats": [l3, l2, l1, l0]}


def sa_ssg(xyz, points, npoint, radius, nsample, mlp, scope, params, gemm='f64'):
    """pointnet_sa_module, group_all=False (pointnet_util.py:86-153 with sample_and_group :21-55):
    channel order [xyz_local, features] (:49).  Returns new_xyz, new_points, idx."""
    new_xyz = O.gather_point(xyz, O.farthest_point_sample(npoint, xyz))
    idx, _ = O.query_ball_point(radius, nsample, xyz, new_xyz)
    g = O.group_point(xyz, idx) - new_xyz[:, :, None, :]
    if points is not None:
        g = np.concatenate([g, O.group_point(points, idx)], -1)
    for j in range(len(mlp)):
        g = _layer(g, params['%s/conv%d' % (scope, j)], gemm)
    return new_xyz, g.max(axis=2), idx


def sa_all(xyz, points, mlp, scope, params, gemm='f64'):
    """pointnet_sa_module, group_all=True (sample_and_group_all, pointnet_util.py:58-83): one group
    of every point, un-centred [xyz, features] (:77), new_xyz = (0,0,0)."""
    b = xyz.shape[0]
    g = (np.concatenate([xyz, points], -1) if points is not None else xyz)[:, None]
    for j in range(len(mlp)):
        g = _layer(g, params['%s/conv%d' % (scope, j)], gemm)
    return np.zeros((b, 1, 3), np.float32), g.max(axis=2)


def _frustum_trunk(xyz0, p0, prefix, params, gemm):
    x1, f1 = sa_msg(xyz0, p0, 128, [0.2, 0.4, 0.8], [32, 64, 128], [[32, 32, 64], [64, 64, 128], [64, 96, 128]],
                    prefix + 'layer1', params, gemm)
    x2, f2 = sa_msg(x1, f1, 32, [0.4, 0.8, 1.6], [64, 64, 128], [[64, 64, 128], [128, 128, 256], [128, 128, 256]],
                    prefix + 'layer2', params, gemm)
    return x1, f1, x2, f2


def frustum_cls_stack(pc, params, gemm='f64'):
    """models/frustum_pointnets_v2.py:127-142 -> (B,512)."""
    xyz0, p0 = np.ascontiguousarray(pc[:, :, :3]), np.ascontiguousarray(pc[:, :, 3:])
    _, _, x2, f2 = _frustum_trunk(xyz0, p0, 'cls_', params, gemm)
    _, f3 = sa_all(x2, f2, [128, 256, 512], 'cls_layer3', params, gemm)
    return f3.reshape(pc.shape[0], -1)


def frustum_seg_stack(pc, one_hot, params, gemm='f64'):
    """models/frustum_pointnets_v2.py:208-234 -> per-point features (B,N,128) and the levels."""
    xyz0, p0 = np.ascontiguousarray(pc[:, :, :3]), np.ascontiguousarray(pc[:, :, 3:])
    x1, f1, x2, f2 = _frustum_trunk(xyz0, p0, '', params, gemm)
    x3, f3 = sa_all(x2, f2, [128, 256, 1024], 'layer3', params, gemm)
    f3 = np.concatenate([f3, one_hot[:, None, :].astype(np.float32)], 2)                 # :226
    g2 = fp(x2, x3, f2, f3, [128, 128], 'fa_layer1', params, gemm)                      # three_nn with m = 1 (:228)
    g1 = fp(x1, x2, f1, g2, [128, 128], 'fa_layer2', params, gemm)
    g0 = fp(xyz0, x1, np.concatenate([xyz0, p0], -1), g1, [128, 128], 'fa_layer3', params, gemm)
    return g0, {"xyz": [xyz0, x1, x2, x3], "sa_feats": [p0, f1, f2, f3], "fp_feats": [g2, g1, g0]}


def frustum_box_stack(obj, params, gemm='f64'):
    """models/frustum_pointnets_v2.py:268-283 (l0_points = None) -> (B,512)."""
    xyz0 = np.ascontiguousarray(obj[:, :, :3])
    p0 = np.ascontiguousarray(obj[:, :, 3:]) if obj.shape[2] > 3 else None
    x1, f1, _ = sa_ssg(xyz0, p0, 128, 0.2, 64, [64, 64, 128], 'ssg-layer1', params, gemm)
    x2, f2, _ = sa_ssg(x1, f1, 32, 0.4, 64, [128, 128, 256], 'ssg-layer2', params, gemm)
    _, f3 = sa_all(x2, f2, [256, 256, 512], 'ssg-layer3', params, gemm)
    return f3.reshape(obj.shape[0], -1), {"xyz": [xyz0, x1, x2], "sa_feats": [p0, f1, f2, f3]}
