"""CPU composition of the RPN SA+FP stack from the oracle ops (TEST INFRASTRUCTURE ONLY).

Follows models/rpn.py:159-192 and models/pointnet_util.py:155-228 of the reference op by
op (no fusion), with the inference-time BN folded like the product does.  Used by
tests/ (end-to-end stack parity) and by bench.py's cpu_baseline leg (the "reference CPU
path" of BASELINE.json: C++ CPU ops + transcribed FPS), never by the product.
`gemm='blas'` swaps the float64 triple loop for a NumPy fp32 matmul so that the timed
baseline is not handicapped by a naive GEMM (the reference would use Eigen/cuDNN there).
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
    return l0, {"xyz": xyzs, "sa_feats": feats}
