"""GPU parity of the backward ops through the C ABI.

Bar: bit-exact against the oracle (which is pinned bit-for-bit to the reference's compiled
CPU sources group_point_grad_cpu / interpolate_grad_cpu, tests/test_oracle.py) -- the HIP
path sums every output element in the serial loop's order.  The reference's own tests
(tf_grouping_op_test.py, tf_interpolate_op_test.py) check the gradient error < 1e-4 on
(1,128,16)/(1,8,..) shapes; test_reference_gradient_tests mirrors them.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from pointrcnn_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.as_tensor(a).to(DEV)


def test_grad_golden_from_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "grad_ref.npz"))
    n = int(g["n"])
    pts = torch.empty((g["idx"].shape[0], n, g["grad_out"].shape[-1]), device=DEV)
    out = ops.group_point_grad(pts, _t(g["idx"]), _t(g["grad_out"])).cpu().numpy()
    assert np.array_equal(out, g["grad_points"])
    m = int(g["m3"])
    pts3 = torch.empty((g["idx3"].shape[0], m, g["grad_out3"].shape[-1]), device=DEV)
    out3 = ops.three_interpolate_grad(pts3, _t(g["idx3"]), _t(g["weight3"]), _t(g["grad_out3"])).cpu().numpy()
    assert np.array_equal(out3, g["grad_points3"])


@pytest.mark.parametrize("b,n,m,ns,c", [(2, 300, 40, 8, 1), (1, 64, 10, 3, 3), (3, 700, 128, 16, 5),
                                        (2, 512, 64, 32, 64), (1, 1000, 100, 16, 96), (2, 200, 30, 4, 260)])
def test_group_point_grad_bit_exact(b, n, m, ns, c):
    rng = np.random.default_rng(b * 1000 + c)
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    go = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    want = O.group_point_grad(np.empty((b, n, c), np.float32), idx, go)
    got = ops.group_point_grad(torch.empty((b, n, c), device=DEV), _t(idx), _t(go))
    assert np.array_equal(got.cpu().numpy(), want)
    again = ops.group_point_grad(torch.empty((b, n, c), device=DEV), _t(idx), _t(go))
    assert torch.equal(got, again)                      # deterministic


def test_group_point_grad_heavy_rows_and_ignored_indices():
    """Segments of 65..1024 entries (LDS ranking), > 1024 entries (ordered rescan), rows nobody
    points at (zeros), and out-of-range indices (ignored, as documented in the header)."""
    rng = np.random.default_rng(5)
    b, n, m, ns, c = 2, 400, 300, 16, 8
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    idx[0, :, 4:] = 7                                   # 3600 entries on one row
    idx[1, :40, :] = 11                                 # 640 entries
    idx[1, 40:46, :] = 12                               # 96 entries
    go = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    want = O.group_point_grad(np.empty((b, n, c), np.float32), idx, go)
    got = ops.group_point_grad(torch.empty((b, n, c), device=DEV), _t(idx), _t(go)).cpu().numpy()
    assert np.array_equal(got, want)
    assert (got[0, 300:] == 0).any()
    bad = idx.copy()
    bad[0, 0, 0] = -1
    bad[1, 2, 3] = n
    ok = bad.copy(); 
    go2 = go.copy(); go2[0, 0, 0] = 0; go2[1, 2, 3] = 0
    ok[0, 0, 0] = 0; ok[1, 2, 3] = 0
    want2 = O.group_point_grad(np.empty((b, n, c), np.float32), ok, go2)
    got2 = ops.group_point_grad(torch.empty((b, n, c), device=DEV), _t(bad), _t(go)).cpu().numpy()
    assert np.array_equal(got2, want2)


@pytest.mark.parametrize("b,n,m,c", [(2, 500, 60, 1), (1, 128, 8, 16), (2, 1024, 256, 128), (1, 300, 3, 7),
                                     (1, 2000, 1, 4)])
def test_three_interpolate_grad_bit_exact(b, n, m, c):
    rng = np.random.default_rng(n + c)
    idx = rng.integers(0, m, (b, n, 3)).astype(np.int32)
    w = rng.random((b, n, 3), dtype=np.float32)
    w /= w.sum(-1, keepdims=True)
    go = rng.standard_normal((b, n, c)).astype(np.float32)
    want = O.three_interpolate_grad(np.empty((b, m, c), np.float32), idx, w, go)
    got = ops.three_interpolate_grad(torch.empty((b, m, c), device=DEV), _t(idx), _t(w), _t(go))
    assert np.array_equal(got.cpu().numpy(), want)


def test_gather_point_grad_bit_exact():
    rng = np.random.default_rng(9)
    b, n, m = 3, 1000, 256
    xyz = rng.standard_normal((b, n, 3)).astype(np.float32)
    idx = ops.farthest_point_sample(m, _t(xyz)).cpu().numpy()            # unique indices
    og = rng.standard_normal((b, m, 3)).astype(np.float32)
    want = O.gather_point_grad(xyz, idx, og)
    got = ops.gather_point_grad(_t(xyz), _t(idx), _t(og)).cpu().numpy()
    assert np.array_equal(got, want)
    dup = rng.integers(0, 20, (b, m)).astype(np.int32)                    # many duplicates: serial order
    want = O.gather_point_grad(xyz, dup, og)
    got = ops.gather_point_grad(_t(xyz), _t(dup), _t(og)).cpu().numpy()
    assert np.array_equal(got, want)


def test_empty_and_bad_arguments():
    z = ops.group_point_grad(torch.empty((2, 50, 4), device=DEV), torch.empty((2, 0, 8), dtype=torch.int32, device=DEV),
                             torch.empty((2, 0, 8, 4), device=DEV))
    assert z.shape == (2, 50, 4) and (z == 0).all()
    with pytest.raises(ValueError):
        ops.group_point_grad(torch.empty((2, 50, 4), device=DEV), torch.zeros((2, 3, 8), dtype=torch.int32, device=DEV),
                             torch.zeros((2, 3, 8, 5), device=DEV))
    with pytest.raises(ValueError):
        ops.three_interpolate_grad(torch.empty((1, 5, 4), device=DEV), torch.zeros((1, 9, 2), dtype=torch.int32, device=DEV),
                                   torch.zeros((1, 9, 3), device=DEV), torch.zeros((1, 9, 4), device=DEV))


def test_reference_gradient_tests():
    """tf_grouping_op_test.py:11-28 and tf_interpolate_op_test.py:11-24: gradient error of
    group_point / three_interpolate w.r.t. points below 1e-4, here through torch.autograd
    against the exact Jacobian (both ops are linear in points)."""
    rng = np.random.default_rng(0)
    points = _t(rng.random((1, 128, 16)).astype(np.float32)).requires_grad_()
    xyz1 = _t(rng.random((1, 128, 3)).astype(np.float32))
    xyz2 = _t(rng.random((1, 8, 3)).astype(np.float32))
    idx, _ = ops.query_ball_point(0.3, 32, xyz1, xyz2)
    grouped = ops.group_point(points, idx)
    assert grouped.requires_grad and grouped.shape == (1, 8, 32, 16)
    g = _t(rng.standard_normal(grouped.shape).astype(np.float32))
    grouped.backward(g)
    exact = torch.zeros((128, 16), dtype=torch.float64, device=DEV)
    exact.index_add_(0, idx.reshape(-1).long(), g.reshape(-1, 16).double())
    assert (points.grad[0].double() - exact).abs().max().item() < 1e-4

    pts = _t(rng.random((1, 8, 16)).astype(np.float32)).requires_grad_()
    dist, i3 = ops.three_nn(xyz1, xyz2)
    w = torch.ones_like(dist) / 3.0
    out = ops.three_interpolate(pts, i3, w)
    assert out.shape == (1, 128, 16)
    g = _t(rng.standard_normal(out.shape).astype(np.float32))
    out.backward(g)
    exact = torch.zeros((8, 16), dtype=torch.float64, device=DEV)
    for t in range(3):
        exact.index_add_(0, i3[0, :, t].long(), g[0].double() * w[0, :, t:t + 1].double())
    assert (pts.grad[0].double() - exact).abs().max().item() < 1e-4

    inp = _t(rng.random((2, 64, 3)).astype(np.float32)).requires_grad_()
    fi = ops.farthest_point_sample(16, inp.detach())
    got = ops.gather_point(inp, fi)
    got.sum().backward()
    want = torch.zeros((2, 64, 3), device=DEV)
    for bi in range(2):
        want[bi, fi[bi].long()] = 1.0
    assert torch.equal(inp.grad, want)


def test_group_point_grad_rpn_level_shape_properties():
    """Full RPN level-2 shape (b16, n4096, m1024, ns32, c96): one cloud against the oracle,
    all clouds through the conservation property sum(grad_points) == sum(grad_out)."""
    torch.manual_seed(0)
    b, n, m, ns, c = 16, 4096, 1024, 32, 96
    idx = torch.randint(0, n, (b, m, ns), dtype=torch.int32, device=DEV)
    go = torch.randn((b, m, ns, c), device=DEV)
    got = ops.group_point_grad(torch.empty((b, n, c), device=DEV), idx, go)
    want0 = O.group_point_grad(np.empty((1, n, c), np.float32), idx[:1].cpu().numpy(), go[:1].cpu().numpy())
    assert np.array_equal(got[:1].cpu().numpy(), want0)
    s_in = go.double().sum(dim=(1, 2))
    s_out = got.double().sum(dim=1)
    assert torch.allclose(s_in, s_out, rtol=0, atol=1e-2)
