"""GPU parity of the kNN grouping ops through the C ABI.

Bar: idx and values bit-exact against the oracle, which equals the reference's compiled
selection_sort_cpu on the golden matrices (tests/test_oracle.py), including the order of
exact ties that the swap sequence produces.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from pointrcnn_amd import ops, pointnet_util as PU, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
    return torch.as_tensor(a).to(DEV)


def test_selection_sort_golden_from_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "selection_sort_ref.npz"))
    for name, d, k in (("ties_k3", "d_ties", 3), ("ties_k16", "d_ties", 16), ("ties_k40", "d_ties", 40),
                       ("free_k10", "d_free", 10)):
        outi, out = ops.select_top_k(k, _t(g[d]))
        assert np.array_equal(outi.cpu().numpy(), g[name + "_idx"]), name
        assert np.array_equal(out.cpu().numpy(), g[name + "_val"]), name


@pytest.mark.parametrize("b,m,n,k,levels", [(2, 7, 100, 5, 4), (1, 3, 1000, 32, 3), (2, 5, 300, 128, 50),
                                            (1, 4, 64, 64, 2), (3, 9, 5000, 16, 0), (1, 2, 130, 1, 1)])
def test_selection_sort_bit_exact(b, m, n, k, levels):
    rng = np.random.default_rng(n + k)
    d = rng.random((b, m, n), dtype=np.float32) if levels == 0 else \
        rng.integers(0, levels, (b, m, n)).astype(np.float32)
    d[0, 0, : min(k, n)] = d.max() + 1                      # everything in front gets displaced
    wi, wv = O.select_top_k(k, d)
    gi, gv = ops.select_top_k(k, _t(d))
    assert np.array_equal(gi.cpu().numpy(), wi)
    assert np.array_equal(gv.cpu().numpy(), wv)


def test_selection_sort_negative_values_and_zero_signs():
    d = np.array([[[0.0, -0.0, 1.0, -2.0, -0.0, 0.0, -2.0, 3.0]]], np.float32)
    wi, wv = O.select_top_k(5, d)
    gi, gv = ops.select_top_k(5, _t(d))
    assert np.array_equal(gi.cpu().numpy(), wi)
    assert np.array_equal(gv.cpu().numpy().view(np.int32), wv.view(np.int32))


@pytest.mark.parametrize("b,n,m,k", [(2, 700, 60, 16), (1, 4096, 128, 32), (2, 5000, 33, 64), (1, 50, 10, 50)])
def test_knn_point_bit_exact(b, n, m, k):
    rng = np.random.default_rng(k)
    base = rng.standard_normal((b, max(n // 2, 1), 3)).astype(np.float32)
    xyz1 = np.concatenate([base, base], 1)[:, :n]           # exact duplicates: tied distances
    if xyz1.shape[1] < n:
        xyz1 = np.concatenate([xyz1, rng.standard_normal((b, n - xyz1.shape[1], 3)).astype(np.float32)], 1)
    xyz2 = xyz1[:, rng.choice(n, m, replace=False)].copy()
    wv, wi = O.knn_point(k, xyz1, xyz2)
    gv, gi = ops.knn_point(k, _t(xyz1), _t(xyz2))
    assert np.array_equal(gi.cpu().numpy(), wi)
    assert np.array_equal(gv.cpu().numpy(), wv)


def test_knn_equals_selection_sort_of_the_matrix_and_errors():
    rng = np.random.default_rng(1)
    xyz1 = _t(rng.standard_normal((2, 900, 3)).astype(np.float32))
    xyz2 = _t(rng.standard_normal((2, 40, 3)).astype(np.float32))
    diff = xyz1.unsqueeze(1) - xyz2.unsqueeze(2)            # (b,m,n,3): tf_grouping.py:65-67
    sq = diff * diff
    dist = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
    outi, out = ops.select_top_k(24, dist.contiguous())
    val, idx = ops.knn_point(24, xyz1, xyz2)
    assert torch.equal(idx, outi[:, :, :24]) and torch.equal(val, out[:, :, :24])
    with pytest.raises(ValueError):
        ops.knn_point(0, xyz1, xyz2)
    with pytest.raises(ValueError):
        ops.knn_point(901, xyz1, xyz2)


def test_sa_module_with_knn_grouping_matches_primitive_composition():
    """pointnet_sa_module(knn=True) (pointnet_util.py:40-41): fused path against the op-by-op
    composition on the same kNN indices."""
    pc = _t(S.kitti_batch(0, 2, 2048))
    xyz, feats = pc[:, :, :3].contiguous(), pc[:, :, 3:].contiguous()
    store = PU.ParamStore(seed=3, device=torch.device(DEV))
    saved = PU.FUSED
    try:
        PU.FUSED = True
        x1, f1, i1 = PU.pointnet_sa_module(xyz, feats, 256, 0.5, 32, [32, 32, 64], None, False, False, None,
                                           'knn_sa', knn=True, params=store)
        PU.FUSED = False
        x0, f0, i0 = PU.pointnet_sa_module(xyz, feats, 256, 0.5, 32, [32, 32, 64], None, False, False, None,
                                           'knn_sa', knn=True, params=store)
    finally:
        PU.FUSED = saved
    assert torch.equal(i0, i1) and torch.equal(x0, x1)
    wv, wi = O.knn_point(32, xyz.cpu().numpy(), x1.cpu().numpy())
    assert np.array_equal(i1.cpu().numpy(), wi)
    assert (f0 - f1).abs().max().item() <= 1e-5 * max(1.0, f0.abs().max().item())
