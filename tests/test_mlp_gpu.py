"""GPU parity: shared-MLP / fused SA / FP layers vs the float64-accumulating oracle.
Tolerances (north_star): 1e-5 relative for fp32, 1e-2 for bf16, checked per element
(tests/_tol.py: relative to the element, with a floor tied to the tensor's largest magnitude)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from pointrcnn_amd import ops, pointnet_util as PU, stacks, synthetic as S

pytestmark = pytest.mark.gpu

from _tol import close as _close_elem


def dev(a, cuda):
    return torch.as_tensor(a).to(cuda)


def npy(t):
    return t.detach().cpu().numpy()


def close(got, ref, prec):
    """Elementwise: |err| <= tol * max(|ref|, floor * max|ref|) (tests/_tol.py)."""
    import inspect
    _close_elem(got, ref, prec, what=inspect.stack()[1].function)


def rand_layer(rng, cin, cout):
    lim = np.sqrt(6.0 / (cin + cout))
    W = rng.uniform(-lim, lim, (cin, cout)).astype(np.float32)
    s = rng.uniform(0.5, 1.5, cout).astype(np.float32)
    t = rng.normal(0, 0.1, cout).astype(np.float32)
    return W, s, t


@pytest.mark.parametrize("prec", [ops.PRECISION_FP32, ops.PRECISION_BF16])
@pytest.mark.parametrize("rows,cin,cout", [(128, 4, 16), (1000, 4, 32), (256, 99, 64), (300, 259, 128),
                                           (512, 515, 256), (130, 196, 256), (64, 256, 384), (257, 1536, 512),
                                           (100, 7, 5), (40, 260, 128), (2048, 16, 16), (96, 33, 196)])
def test_mlp_layer(cuda, prec, rows, cin, cout):
    rng = np.random.default_rng(rows + cin + cout)
    x = rng.standard_normal((rows, cin)).astype(np.float32)
    W, s, t = rand_layer(rng, cin, cout)
    for relu in (True, False):
        got = npy(ops.mlp_layer(dev(x, cuda), dev(W, cuda), dev(s, cuda), dev(t, cuda), relu=relu, precision=prec))
        close(got, O.mlp_layer(x, W, s, t, relu), prec)


def test_mlp_layer_is_batch_invariant(cuda):
    """The launch heuristics (tile width, K-chunk depth) depend on the number of rows; the value of a
    row must not: the first 4096 rows of a 65536-row launch equal a 4096-row launch bit for bit."""
    g = torch.Generator(device='cpu').manual_seed(11)
    x = torch.randn((65536, 512), generator=g).to(cuda)
    W = (torch.randn((512, 512), generator=g) * 0.05).to(cuda)
    s = torch.rand((512,), generator=g).add_(0.5).to(cuda); t = torch.randn((512,), generator=g).to(cuda)
    big = ops.mlp_layer(x, W, s, t)
    small = ops.mlp_layer(x[:4096].contiguous(), W, s, t)
    tiny = ops.mlp_layer(x[:100].contiguous(), W, s, t)
    assert torch.equal(big[:4096], small) and torch.equal(big[:100], tiny)


@pytest.mark.parametrize("prec", [ops.PRECISION_FP32, ops.PRECISION_BF16])
@pytest.mark.parametrize("groups,ns,cin,cout", [(40, 16, 16, 32), (9, 32, 64, 128), (5, 64, 64, 128),
                                                (3, 128, 96, 128), (7, 32, 256, 512), (4, 48, 20, 24),
                                                (2, 1024, 12, 40)])
def test_mlp_layer_fused_pool(cuda, prec, groups, ns, cin, cout):
    rng = np.random.default_rng(groups * ns + cin)
    x = rng.standard_normal((groups * ns, cin)).astype(np.float32)
    W, s, t = rand_layer(rng, cin, cout)
    got = npy(ops.mlp_layer(dev(x, cuda), dev(W, cuda), dev(s, cuda), dev(t, cuda), pool=ns, precision=prec))
    ref = O.max_pool(O.mlp_layer(x, W, s, t).reshape(groups, ns, cout), ns)
    assert got.shape == (groups, cout)
    close(got, ref, prec)


@pytest.mark.parametrize("prec", [ops.PRECISION_FP32, ops.PRECISION_BF16])
@pytest.mark.parametrize("c,xyz_first,ns,cout", [(1, False, 16, 16), (96, False, 32, 64), (0, True, 64, 64),
                                                 (7, True, 16, 40), (256, False, 16, 128), (5, False, 32, 32)])
def test_sa_group_mlp_layer_equals_materialised_path(cuda, prec, c, xyz_first, ns, cout):
    rng = np.random.default_rng(c + ns)
    b, n, m = 2, 700, 50
    xyz = rng.random((b, n, 3), dtype=np.float32)
    feats = rng.standard_normal((b, n, c)).astype(np.float32) if c else None
    fi = O.farthest_point_sample(m, xyz)
    new_xyz = O.gather_point(xyz, fi)
    idx, _ = O.query_ball_point(0.25, ns, xyz, new_xyz)
    W, s, t = rand_layer(rng, c + 3, cout)
    gx = O.group_point(xyz, idx) - new_xyz[:, :, None, :]
    if c:
        gf = O.group_point(feats, idx)
        rows = np.concatenate([gx, gf] if xyz_first else [gf, gx], -1)
    else:
        rows = gx
    ref = O.mlp_layer(rows, W, s, t)
    args = (dev(xyz, cuda), None if feats is None else dev(feats, cuda), dev(new_xyz, cuda), dev(idx, cuda),
            dev(W, cuda), dev(s, cuda), dev(t, cuda))
    got = npy(ops.sa_group_mlp_layer(*args, xyz_first=xyz_first, precision=prec))
    assert got.shape == (b, m, ns, cout)
    close(got, ref, prec)
    got = npy(ops.sa_group_mlp_layer(*args, xyz_first=xyz_first, pool=ns, precision=prec))
    close(got, O.max_pool(ref, ns), prec)


@pytest.mark.parametrize("prec", [ops.PRECISION_FP32, ops.PRECISION_BF16])
@pytest.mark.parametrize("c,ns,couts", [(96, 16, (64, 64)), (256, 32, (128, 128)), (5, 16, (40,)), (512, 32, (256, 7))])
def test_sa_group_add_layer_equals_concat_layer(cuda, prec, c, ns, couts):
    """The hoisted form  (f W_f)[idx] + (xyz[idx]-c) W_x  against the oracle's concat convolution
    [f[idx], xyz[idx]-c] W, scales sharing one per-point table (tolerance tier of the precision)."""
    rng = np.random.default_rng(c + ns)
    b, n, m = 2, 600, 37                                    # ragged last tile, m*ns not a multiple of 128
    xyz = (rng.random((b, n, 3), dtype=np.float32) * np.float32(40.0))   # KITTI-sized coordinates
    feats = rng.standard_normal((b, n, c)).astype(np.float32)
    fi = O.farthest_point_sample(m, xyz)
    new_xyz = O.gather_point(xyz, fi)
    idx, _ = O.query_ball_point(6.0, ns, xyz, new_xyz)
    layers = [rand_layer(rng, c + 3, co) for co in couts]
    wf = np.concatenate([W[:c] for W, _, _ in layers], 1)
    ctab = wf.shape[1]
    table = ops.mlp_layer(dev(feats.reshape(b * n, c), cuda), dev(wf, cuda), torch.ones(ctab, device=cuda),
                          torch.zeros(ctab, device=cuda), relu=False, precision=prec).reshape(b, n, ctab)
    rows = np.concatenate([O.group_point(feats, idx), O.group_point(xyz, idx) - new_xyz[:, :, None, :]], -1)
    col = 0
    for (W, s, t), co in zip(layers, couts):
        ref = O.mlp_layer(rows, W, s, t)
        args = (dev(xyz, cuda), dev(new_xyz, cuda), dev(idx, cuda), table, col, co, dev(W[c:].copy(), cuda),
                dev(s, cuda), dev(t, cuda))
        got = npy(ops.sa_group_add_layer(*args, precision=prec))
        assert got.shape == (b, m, ns, co)
        close(got, ref, prec)
        close(npy(ops.sa_group_add_layer(*args, pool=ns, precision=prec)), O.max_pool(ref, ns), prec)
        col += co
    with pytest.raises(ValueError):
        ops.sa_group_add_layer(dev(xyz, cuda), dev(new_xyz, cuda), dev(idx, cuda), table, ctab - 1, 8,
                               torch.zeros(3, 8, device=cuda), torch.ones(8, device=cuda), torch.zeros(8, device=cuda))


@pytest.mark.parametrize("c,xyz_first,ns,mlp", [(1, False, 16, [16, 16, 32]), (1, False, 32, [32, 32, 64]),
                                                 (0, True, 16, [20, 8, 100]), (4, True, 32, [32, 32, 128]),
                                                 (5, False, 16, [7, 32, 33])])
def test_sa_chain_kernel_equals_oracle_composition(cuda, c, xyz_first, ns, mlp):
    """group + 3 layers + max-pool in one launch (prcnn_sa_group_mlp3_pool) vs the oracle's
    op-by-op composition; odd group counts exercise the half-filled last tile."""
    rng = np.random.default_rng(c * 7 + ns)
    b, n, m = 2, 900, 37
    xyz = rng.random((b, n, 3), dtype=np.float32)
    feats = rng.standard_normal((b, n, c)).astype(np.float32) if c else None
    new_xyz = O.gather_point(xyz, O.farthest_point_sample(m, xyz))
    idx, _ = O.query_ball_point(0.3, ns, xyz, new_xyz)
    layers, cin = [], c + 3
    for co in mlp:
        layers.append(rand_layer(rng, cin, co)); cin = co
    gx = O.group_point(xyz, idx) - new_xyz[:, :, None, :]
    g = gx if not c else np.concatenate([gx, O.group_point(feats, idx)] if xyz_first else [O.group_point(feats, idx), gx], -1)
    for L in layers:
        g = O.mlp_layer(g, *L)
    ref = O.max_pool(g, ns)
    assert ops.sa_chain_supported(c, mlp, ns, ops.PRECISION_FP32)
    got = ops.sa_group_mlp3_pool(dev(xyz, cuda), None if feats is None else dev(feats, cuda), dev(new_xyz, cuda),
                                 dev(idx, cuda), [tuple(dev(t, cuda) for t in L) for L in layers], xyz_first=xyz_first)
    close(npy(got), ref, ops.PRECISION_FP32)
    # writing into a column slice of a wider buffer leaves the rest untouched
    buf = torch.full((b, m, mlp[-1] + 5), -3.0, device=cuda)
    ops.sa_group_mlp3_pool(dev(xyz, cuda), None if feats is None else dev(feats, cuda), dev(new_xyz, cuda),
                           dev(idx, cuda), [tuple(dev(t, cuda) for t in L) for L in layers], xyz_first=xyz_first,
                           out=buf, out_col=2)
    gb = npy(buf)
    close(gb[:, :, 2:2 + mlp[-1]], ref, ops.PRECISION_FP32)
    assert (gb[:, :, :2] == -3.0).all() and (gb[:, :, 2 + mlp[-1]:] == -3.0).all()


def _oracle_sa_msg(xyz, points, npoint, radii, nsamples, mlps, layers):
    new_xyz = O.gather_point(xyz, O.farthest_point_sample(npoint, xyz))
    outs = []
    for i, (r, ns) in enumerate(zip(radii, nsamples)):
        idx, _ = O.query_ball_point(r, ns, xyz, new_xyz)
        g = O.group_point(xyz, idx) - new_xyz[:, :, None, :]
        if points is not None:
            g = np.concatenate([O.group_point(points, idx), g], -1)      # [features, xyz] (:183)
        for (W, s, t) in layers[i]:
            g = O.mlp_layer(g, W, s, t)
        outs.append(O.max_pool(g, ns))
    return new_xyz, np.concatenate(outs, -1)


@pytest.mark.parametrize("fused", [True, False])
def test_sa_module_msg_matches_oracle_composition(cuda, fused):
    rng = np.random.default_rng(17)
    pc = S.kitti_batch(20, 2, 2048)
    xyz, pts = pc[:, :, :3].copy(), pc[:, :, 3:].copy()
    radii, nsamples, mlps = [0.5, 1.0], [16, 32], [[16, 16, 32], [32, 32, 64]]
    store = PU.ParamStore(seed=1, device=cuda, randomize_bn=True)
    PU.FUSED = fused
    try:
        new_xyz, feats = PU.pointnet_sa_module_msg(dev(xyz, cuda), dev(pts, cuda), 256, radii, nsamples, mlps,
                                                   False, None, scope='layer1', params=store)
    finally:
        PU.FUSED = True
    layers = [[tuple(npy(t) for t in store['layer1/conv%d_%d' % (i, j)]) for j in range(3)] for i in range(2)]
    oxyz, ofeats = _oracle_sa_msg(xyz, pts, 256, radii, nsamples, mlps, layers)
    assert np.array_equal(npy(new_xyz), oxyz)
    close(npy(feats), ofeats, ops.PRECISION_FP32)


@pytest.mark.parametrize("c_feat,npoint,mlps", [(96, 333, [[64, 64, 128], [64, 96, 128]]),     # RPN level 2 (rpn.py:170-173)
                                                 (7, 50, [[40, 90, 100], [64, 33, 128]])])      # odd widths, ragged tiles
def test_sa_chain2_hoisted_scale_matches_oracle_and_layerwise(cuda, c_feat, npoint, mlps):
    """prcnn_sa_group_add_mlp2_pool (table gather + 2 MFMA layers + max-pool in one launch) through
    pointnet_sa_module_msg: against the oracle composition and against the layer-by-layer path."""
    rng = np.random.default_rng(23)
    pc = S.kitti_batch(30, 2, 2048)
    xyz = pc[:, :, :3].copy()
    pts = rng.standard_normal((2, 2048, c_feat)).astype(np.float32)
    radii, nsamples = [0.5, 1.0], [16, 32]
    store = PU.ParamStore(seed=4, device=cuda, randomize_bn=True)
    calls = []
    real = ops.sa_group_add_mlp2_pool
    ops.sa_group_add_mlp2_pool = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        new_xyz, feats = PU.pointnet_sa_module_msg(dev(xyz, cuda), dev(pts, cuda), npoint, radii, nsamples, mlps,
                                                   False, None, scope='layer2', params=store)
    finally:
        ops.sa_group_add_mlp2_pool = real
    assert len(calls) == 2                                         # both scales took the one-launch path
    ops.SA_CHAIN2 = False
    try:
        _, feats_lw = PU.pointnet_sa_module_msg(dev(xyz, cuda), dev(pts, cuda), npoint, radii, nsamples, mlps,
                                                False, None, scope='layer2', params=store)
    finally:
        ops.SA_CHAIN2 = True
    layers = [[tuple(npy(t) for t in store['layer2/conv%d_%d' % (i, j)]) for j in range(3)] for i in range(2)]
    oxyz, ofeats = _oracle_sa_msg(xyz, pts, npoint, radii, nsamples, mlps, layers)
    assert np.array_equal(npy(new_xyz), oxyz)
    close(npy(feats), ofeats, ops.PRECISION_FP32)
    close(npy(feats), npy(feats_lw), ops.PRECISION_FP32)


@pytest.mark.parametrize("radii", [[0.1, 0.5], [0.8, 2.0], [3.0, 8.0]])     # sparse, mixed and full balls
def test_sa_dedup_classes_are_bit_identical(cuda, radii):
    """The chain kernels with class lists, and the layer-by-layer path over compacted rows (padded
    copies of a ball's first hit skipped), against the plain layout: same bits, nsample 16 and 32,
    every class populated."""
    rng = np.random.default_rng(31)
    pc = S.kitti_batch(60, 2, 4096)
    xyz = dev(pc[:, :, :3].copy(), cuda)
    new_xyz = ops.gather_point(xyz, ops.farthest_point_sample(777, xyz))
    seen = np.zeros(6, np.int64)
    for (idx, cnt), ns in zip(ops.query_ball_point_msg(radii, [16, 32], xyz, new_xyz), (16, 32)):
        lst, count = ops.ball_classes(cnt)
        c = npy(count)[:6]
        seen += c
        assert c.sum() == cnt.numel()
        cn = npy(cnt).reshape(-1)
        bounds = [1, 2, 4, 8, 16, 32]
        got = np.concatenate([npy(lst)[k, :c[k]] for k in range(6)])
        assert np.array_equal(np.sort(got), np.arange(cn.size))                  # a partition of the centroids
        for k in range(6):
            members = cn[npy(lst)[k, :c[k]]]
            assert (members <= bounds[k]).all() and (k == 0 or (members > bounds[k - 1]).all())
    store = PU.ParamStore(seed=5, device=cuda, randomize_bn=True)
    for c_feat, mlps, scope in ((1, [[16, 16, 32], [32, 32, 64]], 'layer1'), (96, [[64, 64, 128], [64, 96, 128]], 'layer2'),
                                (256, [[128, 196, 256], [128, 196, 256]], 'layer3')):       # compacted rows, layer by layer
        pts = dev(rng.standard_normal((2, 4096, c_feat)).astype(np.float32), cuda)
        outs = []
        for dedup in (True, False):
            PU.SA_DEDUP = dedup
            try:
                outs.append(PU.pointnet_sa_module_msg(xyz, pts, 777, radii, [16, 32], mlps, False, None, scope=scope,
                                                      params=store, new_xyz=new_xyz)[1])
            finally:
                PU.SA_DEDUP = True
        assert torch.equal(outs[0], outs[1])
    if radii[0] == 0.8:
        assert (seen > 0).all()                                                  # the mixed case exercises every class


@pytest.mark.parametrize("fused", [True, False])
def test_sa_module_ssg_group_all_and_fp_match_oracle(cuda, fused):
    fr = S.frustum_batch(3, 4, 512, channels=3)
    xyz = fr[:, :, :3].copy()
    store = PU.ParamStore(seed=2, device=cuda, randomize_bn=True)
    PU.FUSED = fused
    try:
        x1, p1, idx1 = PU.pointnet_sa_module(dev(xyz, cuda), None, 128, 0.2, 64, [64, 64, 128], None, False,
                                             False, None, scope='ssg-layer1', params=store)
        x2, p2, _ = PU.pointnet_sa_module(x1, p1, 32, 0.4, 64, [128, 128, 256], None, False,
                                          False, None, scope='ssg-layer2', params=store)
        x3, p3, _ = PU.pointnet_sa_module(x2, p2, None, None, None, [256, 256, 512], None, True,
                                          False, None, scope='ssg-layer3', params=store)
        f2 = PU.pointnet_fp_module(x2, x3, p2, p3, [128, 128], False, None, scope='fa1', params=store)
        f1 = PU.pointnet_fp_module(x1, x2, p1, f2, [128], False, None, scope='fa2', params=store)
    finally:
        PU.FUSED = True
    L = lambda name: tuple(npy(t) for t in store[name])
    # oracle composition (SSG channel order [xyz, features], pointnet_util.py:49; group_all un-centred, :77)
    ox1 = O.gather_point(xyz, O.farthest_point_sample(128, xyz))
    i1, _ = O.query_ball_point(0.2, 64, xyz, ox1)
    assert np.array_equal(npy(idx1), i1)
    g = O.group_point(xyz, i1) - ox1[:, :, None, :]
    for j in range(3):
        g = O.mlp_layer(g, *L('ssg-layer1/conv%d' % j))
    op1 = O.max_pool(g, 64)
    ox2 = O.gather_point(ox1, O.farthest_point_sample(32, ox1))
    i2, _ = O.query_ball_point(0.4, 64, ox1, ox2)
    g = np.concatenate([O.group_point(ox1, i2) - ox2[:, :, None, :], O.group_point(op1, i2)], -1)
    for j in range(3):
        g = O.mlp_layer(g, *L('ssg-layer2/conv%d' % j))
    op2 = O.max_pool(g, 64)
    g = np.concatenate([ox2, op2], -1)[:, None]
    for j in range(3):
        g = O.mlp_layer(g, *L('ssg-layer3/conv%d' % j))
    op3 = O.max_pool(g, 32)
    ox3 = np.zeros((4, 1, 3), np.float32)
    assert np.array_equal(npy(x1), ox1) and np.array_equal(npy(x2), ox2) and np.array_equal(npy(x3), ox3)
    close(npy(p1), op1, 0); close(npy(p2), op2, 0); close(npy(p3), op3, 0)
    d, i3 = O.three_nn(ox2, ox3)                                  # m = 1 (frustum_pointnets_v2.py:228)
    h = np.concatenate([O.three_interpolate(op3, i3, O.fp_weights(d)), op2], -1)
    for j in range(2):
        h = O.mlp_layer(h, *L('fa1/conv_%d' % j))
    close(npy(f2), h, 0)
    d, i3 = O.three_nn(ox1, ox2)
    h2 = np.concatenate([O.three_interpolate(h, i3, O.fp_weights(d)), op1], -1)
    h2 = O.mlp_layer(h2, *L('fa2/conv_0'))
    close(npy(f1), h2, 0)


def test_rpn_stack_runs_and_fused_equals_unfused(cuda):
    pc = dev(S.kitti_batch(40, 1), cuda)
    store = PU.ParamStore(seed=7, device=cuda)
    out, aux = stacks.rpn_sa_fp_stack(pc, params=store)
    assert out.shape == (1, 16384, 128)
    assert [x.shape[1] for x in aux["xyz"]] == [16384, 4096, 1024, 256, 64]
    assert [f.shape[2] for f in aux["sa_feats"][1:]] == [96, 256, 512, 1024]
    PU.FUSED = False
    try:
        out2, _ = stacks.rpn_sa_fp_stack(pc, params=store)
    finally:
        PU.FUSED = True
    close(npy(out), npy(out2), ops.PRECISION_FP32)
    # (both paths and the bf16 tier against the ORACLE at this size: tests/test_stack_parity_gpu.py)


@pytest.mark.parametrize("split", [True, False])
def test_rpn_stack_with_precomputed_geometry_is_identical(cuda, split):
    """Centroids, ball queries and three_nn + weights computed beforehand (stacks.rpn_geometry, on a
    batch of 5 clouds -- above ops.FP_TWO_STEP_MAX_QUERIES at the last level, so the in-line stack
    takes the fused affine kernel there -- consumed cloud by cloud) give bit-identical features;
    the search-free affine interpolation equals the fused search + interpolation kernel."""
    pc = dev(S.kitti_batch(50, 5), cuda)
    store = PU.ParamStore(seed=7, device=cuda, randomize_bn=True)
    PU.FP_SPLIT = split
    try:
        ref, aux = stacks.rpn_sa_fp_stack(pc, params=store)
        xyz = pc[:, :, 0:3].contiguous()
        cents, _ = stacks.sample_centroid_chain(xyz, [lv[0] for lv in stacks.RPN_SA])
        geo = stacks.rpn_geometry(xyz, cents)
        assert len(stacks.geometry_tensors(geo)) == 2 * 8 + 2 * 4
        both = stacks.geometry_clone(geo)
        for i in (0, 4):
            out, _ = stacks.rpn_sa_fp_stack(pc[i:i + 1], params=store, centroids=[c[i:i + 1] for c in cents],
                                            geometry=stacks.geometry_slice(both, i, i + 1))
            assert torch.equal(out, ref[i:i + 1])
    finally:
        PU.FP_SPLIT = True
    # the search-free affine epilogue on its own: p1 with 1 and 4 channels, and none
    b, n, m, c = 2, 5000, 700, 64
    g = torch.Generator(device='cpu').manual_seed(3)
    x1 = torch.rand((b, n, 3), generator=g).to(cuda); x2 = torch.rand((b, m, 3), generator=g).to(cuda)
    p2 = torch.randn((b, m, c), generator=g).to(cuda); t = torch.randn((c,), generator=g).to(cuda)
    dist, idx = ops.three_nn(x1, x2)
    w = ops.fp_weights(dist)
    for c1 in (0, 1, 4):
        p1 = torch.randn((b, n, c1), generator=g).to(cuda) if c1 else None
        u = torch.randn((c1, c), generator=g).to(cuda) if c1 else None
        got = ops.three_interpolate_affine(p2, idx, w, p1, u, t, relu=True)
        want = ops.three_interpolate(p2, idx, w) + t
        if c1:
            # epilogue order: t + p1[0] u[0] + p1[1] u[1] + ... (unfused), then added to the interpolation
            acc = t.expand(b, n, c).clone()
            for k in range(c1):
                acc = acc + p1[:, :, k:k + 1] * u[k].view(1, 1, c)
            want = ops.three_interpolate(p2, idx, w) + acc
        assert torch.equal(got, torch.relu(want))


def test_frustum_stacks_run(cuda):
    store = PU.ParamStore(seed=9, device=cuda)
    fr = dev(S.frustum_batch(1, 32, 1024), cuda)
    assert stacks.frustum_cls_stack(fr, params=store).shape == (32, 512)
    onehot = torch.zeros((32, 4), device=cuda); onehot[:, 1] = 1
    assert stacks.frustum_seg_stack(fr, onehot, params=store).shape == (32, 1024, 128)
    obj = dev(S.frustum_batch(2, 32, 512, channels=3), cuda)
    assert stacks.frustum_box_stack(obj, params=store).shape == (32, 512)
