"""PointNet++ set-abstraction / feature-propagation layers on the gfx950 ops.

Host-side mirror of the reference's models/pointnet_util.py with the same function
names and argument lists:
  sample_and_group            :21-55      sample_and_group_all        :58-83
  pointnet_sa_module          :86-153     pointnet_sa_module_msg      :155-195
  pointnet_fp_module          :198-228
Inference only (is_training must be False): tf_util.conv2d's conv + bias + batch_norm
(models/tf_util.py:120-185,512-531, eps 1e-3) is folded into one (W, scale, shift)
triple per layer, looked up in a ParamStore under the TF variable-scope name the
reference would have used ('<scope>/conv0', '<scope>/conv1_2', '<scope>/conv_0').

Two execution modes, same results (tests/test_pointnet_util.py):
  FUSED = True   grouping is fused into the first shared-MLP layer and the max-pool into
                 the last one; FP uses the fused three_nn+interpolate kernel writing
                 straight into the concat buffer.  The (b,m,nsample,C) tensor never exists.
  FUSED = False  the reference's op-by-op composition (group_point, concat, ...).
Options of the reference that no caller uses raise NotImplementedError (pooling other
than 'max', mlp2, knn, use_nchw).
"""
import math

import os

import torch

from . import ops

FUSED = True
PRECISION = ops.PRECISION_FP32


class ParamStore(dict):
    """scope name -> (W (cin,cout), scale (cout,), shift (cout,)) float32 GPU tensors."""

    def __init__(self, seed=7, device="cuda", randomize_bn=False):
        super().__init__()
        self.gen = torch.Generator(device="cpu")
        self.gen.manual_seed(seed)
        self.device = device
        self.randomize_bn = randomize_bn

    def zeros(self, n):
        key = '__zeros_%d' % n
        if key not in self:
            self[key] = torch.zeros(n, dtype=torch.float32, device=self.device)
        return self[key]

    def ones(self, n):
        key = '__ones_%d' % n
        if key not in self:
            self[key] = torch.ones(n, dtype=torch.float32, device=self.device)
        return self[key]

    def derived(self, key, make):
        """Tensors derived from stored layers (weight slices / concatenations), built once."""
        if key not in self:
            self[key] = make()
        return self[key]

    def layer(self, name, cin, cout, bn=True):
        if name not in self:
            lim = math.sqrt(6.0 / (cin + cout))                     # xavier_initializer
            W = (torch.rand((cin, cout), generator=self.gen) * 2 - 1) * lim
            if self.randomize_bn:
                gamma = torch.rand(cout, generator=self.gen) + 0.5
                beta = torch.randn(cout, generator=self.gen) * 0.1
                mean = torch.randn(cout, generator=self.gen) * 0.1
                var = torch.rand(cout, generator=self.gen) + 0.5
                bias = torch.randn(cout, generator=self.gen) * 0.05
            else:
                gamma, beta = torch.ones(cout), torch.zeros(cout)
                mean, var, bias = torch.zeros(cout), torch.ones(cout), torch.zeros(cout)
            self[name] = fold_bn(W, bias, gamma, beta, mean, var, bn=bn, device=self.device)
        W, s, t = self[name]
        if W.shape != (cin, cout):
            raise ValueError("%s: stored weight %s does not match (%d,%d)" % (name, tuple(W.shape), cin, cout))
        return W, s, t


SA_DEDUP = os.environ.get("PRCNN_SA_DEDUP", "1") != "0"      # chain kernels skip the padded copies of a ball's first hit (ops.ball_classes; same bits)
SA_SPLIT = True      # hoist the feature half of the first MSG layer out of the grouping (pointnet_sa_module_msg)
FP_SPLIT = True      # split the first FP convolution around the interpolation (pointnet_fp_module)


def fold_bn(W, bias, gamma, beta, mean, var, eps=1e-3, bn=True, device="cuda"):
    """relu(BN(x.W + bias)) == relu(scale * (x.W) + shift)   (tf_util.py:170-184,:527)."""
    if bn:
        scale = gamma / torch.sqrt(var + eps)
        shift = beta + (bias - mean) * scale
    else:
        scale = torch.ones_like(bias)
        shift = bias.clone()
    f = lambda t: t.to(device=device, dtype=torch.float32).contiguous()
    return f(W), f(scale), f(shift)


_PARAMS = None


def set_params(store):
    global _PARAMS
    _PARAMS = store
    return store


def get_params():
    global _PARAMS
    if _PARAMS is None:
        _PARAMS = ParamStore()
    return _PARAMS


def _check_inference(is_training):
    if bool(is_training):
        raise NotImplementedError("inference only: is_training must be False (gradients are a 'next' row)")


def sample_and_group(npoint, radius, nsample, xyz, points, knn=False, use_xyz=True):
    """models/pointnet_util.py:21-55 -> new_xyz, new_points, idx, grouped_xyz."""
    new_xyz = ops.gather_point(xyz, ops.farthest_point_sample(npoint, xyz))
    if knn:
        _, idx = ops.knn_point(nsample, xyz, new_xyz)                # pointnet_util.py:40-41
    else:
        idx, pts_cnt = ops.query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = ops.group_point(xyz, idx)
    grouped_xyz = grouped_xyz - new_xyz.unsqueeze(2)                 # translation normalisation
    if points is not None:
        grouped_points = ops.group_point(points, idx)
        new_points = torch.cat([grouped_xyz, grouped_points], dim=-1) if use_xyz else grouped_points
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def sample_and_group_all(xyz, points, use_xyz=True):
    """models/pointnet_util.py:58-83: one group holding every point, centroid (0,0,0), xyz un-centred."""
    b, n, _ = xyz.shape
    new_xyz = torch.zeros((b, 1, 3), dtype=torch.float32, device=xyz.device)
    idx = torch.arange(n, dtype=torch.int32, device=xyz.device).view(1, 1, n).repeat(b, 1, 1)
    grouped_xyz = xyz.reshape(b, 1, n, 3)
    if points is not None:
        new_points = torch.cat([xyz, points], dim=2) if use_xyz else points
        new_points = new_points.unsqueeze(1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def _mlp_chain(x, names_dims, store, bn, pool, precision):
    """x (..., cin) through the layers; the max-pool is fused into the last layer."""
    for li, (name, cin, cout) in enumerate(names_dims):
        W, s, t = store.layer(name, cin, cout, bn)
        last = li == len(names_dims) - 1
        x = ops.mlp_layer(x, W, s, t, relu=True, pool=pool if last else 1, precision=precision)
    return x


def pointnet_sa_module(xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, is_training,
                       bn_decay, scope, bn=True, pooling='max', knn=False, use_xyz=True,
                       use_nchw=False, params=None, precision=None):
    """models/pointnet_util.py:86-153 -> new_xyz (b,npoint,3), new_points (b,npoint,mlp[-1]), idx."""
    _check_inference(is_training)
    if pooling != 'max' or mlp2 is not None or use_nchw:
        raise NotImplementedError("only pooling='max', mlp2=None, NHWC are used by the reference")
    store = params if params is not None else get_params()
    precision = PRECISION if precision is None else precision
    b = xyz.shape[0]
    if group_all:
        nsample = xyz.shape[1]
        new_xyz, new_points, idx, _ = sample_and_group_all(xyz, points, use_xyz)
        cin = new_points.shape[-1]
        dims, c = [], cin
        for i, co in enumerate(mlp):
            dims.append(('%s/conv%d' % (scope, i), c, co)); c = co
        out = _mlp_chain(new_points.reshape(b * nsample, cin), dims, store, bn, nsample, precision)
        return new_xyz, out.reshape(b, 1, mlp[-1]), idx
    c_feat = 0 if points is None else points.shape[2]
    if FUSED and (use_xyz or points is None) and nsample % 16 == 0 and 128 % nsample == 0:
        new_xyz = ops.gather_point(xyz, ops.farthest_point_sample(npoint, xyz))
        if knn:
            _, idx = ops.knn_point(nsample, xyz, new_xyz)            # pointnet_util.py:40-41
        else:
            idx, pts_cnt = ops.query_ball_point(radius, nsample, xyz, new_xyz)
        if ops.sa_chain_supported(c_feat, mlp, nsample, precision):
            layers = [store.layer('%s/conv%d' % (scope, j), ci, co, bn)
                      for j, (ci, co) in enumerate(zip([c_feat + 3] + list(mlp[:-1]), mlp))]
            x = ops.sa_group_mlp3_pool(xyz, points, new_xyz, idx, layers, xyz_first=True,
                                       classes=ops.ball_classes(pts_cnt) if SA_DEDUP and not knn else None)
            return new_xyz, x, idx
        W, s, t = store.layer('%s/conv0' % scope, c_feat + 3, mlp[0], bn)
        x = ops.sa_group_mlp_layer(xyz, points, new_xyz, idx, W, s, t, xyz_first=True, relu=True,
                                   pool=nsample if len(mlp) == 1 else 1, precision=precision)
        dims, c = [], mlp[0]
        for i, co in enumerate(mlp[1:], 1):
            dims.append(('%s/conv%d' % (scope, i), c, co)); c = co
        if dims:
            x = _mlp_chain(x.reshape(-1, mlp[0]), dims, store, bn, nsample, precision)
        return new_xyz, x.reshape(b, npoint, mlp[-1]), idx
    new_xyz, new_points, idx, _ = sample_and_group(npoint, radius, nsample, xyz, points, knn, use_xyz)
    cin = new_points.shape[-1]
    dims, c = [], cin
    for i, co in enumerate(mlp):
        dims.append(('%s/conv%d' % (scope, i), c, co)); c = co
    out = _mlp_chain(new_points.reshape(-1, cin), dims, store, bn, nsample, precision)
    return new_xyz, out.reshape(b, npoint, mlp[-1]), idx


def pointnet_sa_module_msg(xyz, points, npoint, radius_list, nsample_list, mlp_list, is_training,
                           bn_decay, scope, bn=True, use_xyz=True, use_nchw=False, params=None,
                           precision=None, new_xyz=None, balls=None, order=None):
    """models/pointnet_util.py:155-195 -> new_xyz (b,npoint,3), new_points (b,npoint,sum mlp[k][-1]).
    Channel order inside each scale is [features, xyz_local] (:183).  order: ops.grid_order(xyz, new_xyz)
    when the caller shares it with another search over the same two sets."""
    _check_inference(is_training)
    if use_nchw:
        raise NotImplementedError("use_nchw is never set by the reference's callers")
    store = params if params is not None else get_params()
    precision = PRECISION if precision is None else precision
    b = xyz.shape[0]
    if new_xyz is None:    # a caller may have sampled the centroids already (on another stream)
        new_xyz = ops.gather_point(xyz, ops.farthest_point_sample(npoint, xyz))
    ctot = sum(m[-1] for m in mlp_list)
    out = torch.empty((b, npoint, ctot), dtype=torch.float32, device=xyz.device)
    col = 0
    c_feat = 0 if points is None else points.shape[2]
    # FUSED: one pass over the points answers the ball queries of every scale; a caller may
    # pass them in (`balls`: [(idx, pts_cnt)] per scale, e.g. answered on another stream -- they
    # depend on coordinates only)
    if balls is None and FUSED:
        balls = ops.query_ball_point_msg(radius_list, nsample_list, xyz, new_xyz, order=order)
    # SA_SPLIT: the first convolution is linear, so its feature half is evaluated once per POINT
    # for all scales together (b*n rows x c_feat x sum c1) instead of once per grouped row
    # (b*m*nsample rows per scale); the grouped layer then gathers those rows and adds the
    # 3-column xyz product:  [f[idx], xyz[idx]-c] W = (f W_f)[idx] + (xyz[idx]-c) W_x.
    # Same value up to fp32 re-association (tolerance tier, like every MLP output).
    split = [FUSED and SA_SPLIT and use_xyz and points is not None and ns_ % 16 == 0 and 128 % ns_ == 0
             and not ops.sa_chain_supported(c_feat, mlp_, ns_, precision)
             for ns_, mlp_ in zip(nsample_list, mlp_list)]
    table, tcol = None, {}
    if any(split):
        names = ['%s/conv%d_0' % (scope, i) for i in range(len(radius_list)) if split[i]]
        for i in range(len(radius_list)):
            if split[i]:
                store.layer('%s/conv%d_0' % (scope, i), c_feat + 3, mlp_list[i][0], bn)
        wf = store.derived('__wf/' + '|'.join(names),
                           lambda: torch.cat([store[nm][0][:c_feat] for nm in names], dim=1).contiguous())
        o = 0
        for i in range(len(radius_list)):
            if split[i]:
                tcol[i] = o
                o += mlp_list[i][0]
        table = ops.mlp_layer(points.reshape(b * points.shape[1], c_feat), wf, store.ones(o), store.zeros(o),
                              relu=False, precision=precision).reshape(b, points.shape[1], o)
    for i in range(len(radius_list)):
        radius, nsample, mlp = radius_list[i], nsample_list[i], mlp_list[i]
        idx, pts_cnt = balls[i] if balls is not None else ops.query_ball_point(radius, nsample, xyz, new_xyz)
        fusable = FUSED and (use_xyz or points is None) and nsample % 16 == 0 and 128 % nsample == 0
        if fusable and ops.sa_chain_supported(c_feat, mlp, nsample, precision):
            # narrow layers: the whole scale (group, 3 layers, max-pool) is one kernel
            layers = [store.layer('%s/conv%d_%d' % (scope, i, j), ci, co, bn)
                      for j, (ci, co) in enumerate(zip([c_feat + 3] + list(mlp[:-1]), mlp))]
            ops.sa_group_mlp3_pool(xyz, points, new_xyz, idx, layers, xyz_first=False, out=out, out_col=col,
                                   classes=ops.ball_classes(pts_cnt) if SA_DEDUP and pts_cnt is not None else None)
            col += mlp[-1]
            continue
        if fusable and split[i] and ops.sa_chain2_supported(mlp, nsample, precision):
            # the whole hoisted scale in one launch: table gather + 2 MFMA layers + max-pool
            name = '%s/conv%d_0' % (scope, i)
            W, s, t = store.layer(name, c_feat + 3, mlp[0], bn)
            wx = store.derived('__wx/' + name, lambda: W[c_feat:].contiguous())
            layers = [(W, s, t)] + [store.layer('%s/conv%d_%d' % (scope, i, j), mlp[j - 1], mlp[j], bn)
                                    for j in (1, 2)]
            ops.sa_group_add_mlp2_pool(xyz, new_xyz, idx, table, tcol[i], wx, layers, out=out, out_col=col,
                                       classes=ops.ball_classes(pts_cnt) if SA_DEDUP and pts_cnt is not None else None)
            col += mlp[-1]
            continue
        # (nsample <= 32 only: with the large, well-filled balls of the Frustum nets -- nsample 64 / 128 on
        # a few hundred object points -- nothing is saved and the small strided grid of the compacted
        # launches loses: 2.2 -> 4.0 ms per Frustum batch measured)
        if fusable and split[i] and SA_DEDUP and pts_cnt is not None and len(mlp) >= 2 and nsample <= 32:
            # layer by layer over the compacted (duplicate-free) rows, then a max per centroid
            name = '%s/conv%d_0' % (scope, i)
            W, s, t = store.layer(name, c_feat + 3, mlp[0], bn)
            wx = store.derived('__wx/' + name, lambda: W[c_feat:].contiguous())
            cr = ops.CompactRows(idx, pts_cnt, xyz.shape[1])
            x = ops.sa_rows_add_layer(cr, xyz, new_xyz, table, tcol[i], mlp[0], wx, s, t, relu=True, precision=precision)
            for j in range(1, len(mlp)):
                Wj, sj, tj = store.layer('%s/conv%d_%d' % (scope, i, j), mlp[j - 1], mlp[j], bn)
                x = ops.mlp_layer_n(cr, x, Wj, sj, tj, relu=True, precision=precision)
            ops.segment_max(cr, x, out, out_col=col)
            col += mlp[-1]
            continue
        if fusable and split[i]:
            name = '%s/conv%d_0' % (scope, i)
            W, s, t = store.layer(name, c_feat + 3, mlp[0], bn)
            wx = store.derived('__wx/' + name, lambda: W[c_feat:].contiguous())
            x = ops.sa_group_add_layer(xyz, new_xyz, idx, table, tcol[i], mlp[0], wx, s, t, relu=True,
                                       pool=nsample if len(mlp) == 1 else 1, precision=precision)
            x = x.reshape(-1, mlp[0])
            start, c = 1, mlp[0]
        elif fusable:
            W, s, t = store.layer('%s/conv%d_0' % (scope, i), c_feat + 3, mlp[0], bn)
            x = ops.sa_group_mlp_layer(xyz, points, new_xyz, idx, W, s, t, xyz_first=False, relu=True,
                                       pool=nsample if len(mlp) == 1 else 1, precision=precision)
            x = x.reshape(-1, mlp[0])
            start, c = 1, mlp[0]
        else:
            grouped_xyz = ops.group_point(xyz, idx) - new_xyz.unsqueeze(2)
            if points is not None:
                grouped_points = ops.group_point(points, idx)
                if use_xyz:
                    grouped_points = torch.cat([grouped_points, grouped_xyz], dim=-1)
            else:
                grouped_points = grouped_xyz
            c = grouped_points.shape[-1]
            x = grouped_points.reshape(-1, c)
            start = 0
        for j in range(start, len(mlp)):
            W, s, t = store.layer('%s/conv%d_%d' % (scope, i, j), c, mlp[j], bn)
            last = j == len(mlp) - 1
            if last:   # pooled result goes straight into its slice of the concat (:194)
                if nsample % 16 == 0 and 128 % nsample == 0:
                    ops.mlp_layer(x, W, s, t, relu=True, pool=nsample, precision=precision,
                                  out=out, out_col=col)
                else:
                    out[:, :, col:col + mlp[j]] = ops.mlp_layer(
                        x, W, s, t, relu=True, pool=nsample, precision=precision).reshape(b, npoint, -1)
            else:
                x = ops.mlp_layer(x, W, s, t, relu=True, precision=precision)
            c = mlp[j]
        if len(mlp) == 1 and fusable:
            out[:, :, col:col + mlp[0]] = x.reshape(b, npoint, -1)
        col += mlp[-1]
    return new_xyz, out


DEBUG_TAP = None


def pointnet_fp_module(xyz1, xyz2, points1, points2, mlp, is_training, bn_decay, scope, bn=True,
                       params=None, precision=None, nn=None, order=None):
    """models/pointnet_util.py:198-228 -> (b, ndataset1, mlp[-1]).  `nn` = (idx, weight) of
    three_nn(xyz1, xyz2) + the inverse-distance weights (:210-214) when the caller has them
    already (they depend on coordinates only).  order: ops.grid_order(xyz1, xyz2) when the caller
    shares it with another search over the same two sets."""
    _check_inference(is_training)
    store = params if params is not None else get_params()
    precision = PRECISION if precision is None else precision
    b, n, _ = xyz1.shape
    c2 = points2.shape[2]
    if FUSED and FP_SPLIT and len(mlp) >= 1:
        # The interpolation is linear, so the wide half of the first convolution runs on the m
        # known rows instead of the n (= 4m) interpolated rows:
        #   relu(s*([interp, p1] W) + t) = relu(interp3((p2 W_a) s) + (p1 W_b) s + t)
        # (concat order [interpolated, points1]: pointnet_util.py:218).  Same value up to fp32
        # re-association (tolerance tier, like every MLP output); FP_SPLIT = False keeps the
        # concat form.
        m = xyz2.shape[1]
        c1 = 0 if points1 is None else points1.shape[2]
        co = mlp[0]
        W, s, t = store.layer('%s/conv_0' % scope, c2 + c1, co, bn)
        y2 = ops.mlp_layer(points2.reshape(b * m, c2), W[:c2], s, store.zeros(co), relu=False, precision=precision)
        if nn is not None:
            idx, weight = nn
            if c1 <= 4:
                name = '%s/conv_0' % scope
                u = store.derived('__fpu/' + name, lambda: (W[c2:] * s.unsqueeze(0)).contiguous()) if c1 else None
                x = ops.three_interpolate_affine(y2.reshape(b, m, co), idx, weight, points1, u, t, relu=True)
            else:
                x = ops.mlp_layer(points1.reshape(b * n, c1), W[c2:], s, t, relu=False, precision=precision)
                ops.three_interpolate_add(y2.reshape(b, m, co), idx, weight, x.reshape(b, n, co), relu=True)
        elif ops.three_nn_interpolate_affine_supported(b, n, c1):
            # narrow points1 (the intensity channel of the last RPN level) or none: its half
            # (p1 W_b) s + t is evaluated in the interpolation epilogue, never stored
            name = '%s/conv_0' % scope
            u = store.derived('__fpu/' + name, lambda: (W[c2:] * s.unsqueeze(0)).contiguous()) if c1 else None
            y2_before = y2.clone() if DEBUG_TAP is not None else None
            x = ops.three_nn_interpolate_affine(xyz1, xyz2, y2.reshape(b, m, co), points1, u, t, relu=True,
                                                order=order)
            if DEBUG_TAP is not None:      # diagnostic: keep the level's intermediates (bench.py PRCNN_BENCH_TAP)
                DEBUG_TAP.append((scope, y2, x, y2_before, y2.clone()))
        else:
            if points1 is not None:
                x = ops.mlp_layer(points1.reshape(b * n, c1), W[c2:], s, t, relu=False, precision=precision)
            else:
                x = t.expand(b * n, co).contiguous()
            ops.three_nn_interpolate_add(xyz1, xyz2, y2.reshape(b, m, co), x.reshape(b, n, co), relu=True)
        c = co
        for i, co in enumerate(mlp[1:], 1):
            W, s, t = store.layer('%s/conv_%d' % (scope, i), c, co, bn)
            x = ops.mlp_layer(x, W, s, t, relu=True, precision=precision)
            c = co
        return x.reshape(b, n, mlp[-1])
    if nn is not None:
        interpolated_points = ops.three_interpolate(points2, nn[0], nn[1])
        new_points1 = (torch.cat([interpolated_points, points1], dim=2)
                       if points1 is not None else interpolated_points)
    elif FUSED:
        c1 = 0 if points1 is None else points1.shape[2]
        new_points1 = torch.empty((b, n, c2 + c1), dtype=torch.float32, device=xyz1.device)
        ops.three_nn_interpolate(xyz1, xyz2, points2, out=new_points1)
        if points1 is not None:
            new_points1[:, :, c2:] = points1
    else:
        dist, idx = ops.three_nn(xyz1, xyz2)
        weight = ops.fp_weights(dist)
        interpolated_points = ops.three_interpolate(points2, idx, weight)
        new_points1 = (torch.cat([interpolated_points, points1], dim=2)
                       if points1 is not None else interpolated_points)
    c = new_points1.shape[2]
    x = new_points1.reshape(b * n, c)
    for i, co in enumerate(mlp):
        W, s, t = store.layer('%s/conv_%d' % (scope, i), c, co, bn)
        x = ops.mlp_layer(x, W, s, t, relu=True, precision=precision)
        c = co
    return x.reshape(b, n, mlp[-1])
