"""The small-footprint sampler (prcnn_farthest_point_sample_shared: cloud in L2, four waves per cloud)
against the CPU oracle and the register-resident sampler: bit-identical indices."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle as O
from pointrcnn_amd import _lib, ops, synthetic as S

pytestmark = pytest.mark.gpu


def dev(a, cuda):
    return torch.as_tensor(a).to(cuda)


def npy(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("n,m", [(1, 3), (2, 2), (100, 100), (127, 40), (128, 128), (129, 64), (700, 90),
                                 (4097, 256), (5000, 300), (8192, 1000), (16383, 700), (16384, 2048)])
def test_shared_sampler_sizes(cuda, n, m):
    """ragged last bucket, empty buckets, more samples than distinct points, heavy duplicates"""
    pts = S.kitti_scene(5000 + n, n, max(64, int(n * 1.15)))[None, :, :3]
    pts = np.concatenate([pts, S.kitti_scene(6000 + n, n, max(64, n // 5))[None, :, :3]], 0)
    got = npy(ops.farthest_point_sample(m, dev(pts, cuda), shared=True))
    assert got.dtype == np.int32
    assert np.array_equal(got, O.farthest_point_sample(m, pts))


def test_shared_sampler_degenerate_clouds_and_goldens(cuda, golden_dir):
    rng = np.random.default_rng(5)
    n, m = 6000, 300
    same = np.tile(np.array([[1.5, -2.0, 30.0]], np.float32), (n, 1))          # every distance 0
    line = np.zeros((n, 3), np.float32); line[:, 2] = rng.permutation(n) * 0.01   # two empty axes
    plane = rng.uniform(-20, 20, (n, 3)).astype(np.float32); plane[:, 1] = 1.65
    lattice = (rng.integers(0, 12, (n, 3)) * 0.5).astype(np.float32)              # masses of exact ties
    pts = np.stack([same, line, plane, lattice])
    got = npy(ops.farthest_point_sample(m, dev(pts, cuda), shared=True))
    assert np.array_equal(got, O.farthest_point_sample(m, pts))
    g = np.load(os.path.join(golden_dir, "fps_transcription.npz"))
    assert np.array_equal(npy(ops.farthest_point_sample(400, dev(g["cloud"], cuda), shared=True)), g["idx_dup"])
    assert np.array_equal(npy(ops.farthest_point_sample(4, dev(g["tie"], cuda), shared=True)), g["idx_tie"])
    assert np.array_equal(npy(ops.farthest_point_sample(100, dev(g["small"], cuda), shared=True)), g["idx_small"])


def test_shared_sampler_full_rpn_level1_batch(cuda):
    """16384 -> 4096 (BASELINE level 1): two scenes against the oracle, a batch of 16 against the
    register-resident sampler; twice in a row (the workspace keeps nothing between calls)."""
    pts = S.kitti_batch(0, 2)[:, :, :3]
    assert np.array_equal(npy(ops.farthest_point_sample(4096, dev(pts, cuda), shared=True)),
                          O.farthest_point_sample(4096, pts))
    big = dev(S.kitti_batch(3, 16)[:, :, :3], cuda)
    want = npy(ops.farthest_point_sample(4096, big, shared=False))
    for _ in range(2):
        assert np.array_equal(npy(ops.farthest_point_sample(4096, big, shared=True)), want)


def test_shared_sampler_argument_checks(cuda):
    L = _lib.lib()
    x = torch.zeros((1, 64, 3), device=cuda)
    out = torch.zeros((1, 8), dtype=torch.int32, device=cuda)
    ws = torch.zeros(L.prcnn_fps_shared_temp_bytes(1, 64), dtype=torch.uint8, device=cuda)
    call = L.prcnn_farthest_point_sample_shared
    assert call(1, 64, 0, x.data_ptr(), ws.data_ptr(), ws.numel(), out.data_ptr(), None) == -1       # npoint > 0
    assert call(1, 0, 8, x.data_ptr(), ws.data_ptr(), ws.numel(), out.data_ptr(), None) == -2
    assert call(1, 64, 8, x.data_ptr(), None, 0, out.data_ptr(), None) == -4                          # workspace
    assert call(1, 64, 8, x.data_ptr(), ws.data_ptr(), ws.numel() - 1, out.data_ptr(), None) == -4
    assert call(1, 20000, 8, x.data_ptr(), ws.data_ptr(), ws.numel(), out.data_ptr(), None) == -5     # n > 16384
    assert L.prcnn_fps_shared_temp_bytes(1, 20000) == 0
    assert call(0, 64, 8, x.data_ptr(), ws.data_ptr(), ws.numel(), out.data_ptr(), None) == 0
    torch.cuda.synchronize()
