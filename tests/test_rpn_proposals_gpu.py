"""GPU parity of the RPN proposal post-processing through the C ABI,
against the NumPy restatement oracle/rpn_numpy.py (parity unpinned: TensorFlow graph ops, no
fixture in the reference).  Bars: indices (argmax picks, NMS keep lists) exact; decoded floats
within 1e-6 relative of the fp32 restatement (expf / sinf / cosf differ in the last ulp)."""
import numpy as np
import pytest
import torch

from oracle import rpn_numpy as R
from pointrcnn_amd import rpn_proposals as P

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _inputs(B, M, seed):
    rng = np.random.default_rng(seed)
    out = rng.standard_normal((B, M, 4 * 12 + 1 + 2 * 12 + 4 * 3)).astype(np.float32)
    xyz = np.stack([rng.uniform(-30, 30, (B, M)), rng.uniform(-1, 2, (B, M)), rng.uniform(2, 60, (B, M))], -1)
    # proposals cluster around few objects, like the RPN's foreground points
    xyz[:, :, 0] = np.round(xyz[:, :, 0] / 6) * 6 + rng.normal(0, 0.3, (B, M))
    xyz[:, :, 2] = np.round(xyz[:, :, 2] / 8) * 8 + rng.normal(0, 0.3, (B, M))
    logits = rng.standard_normal((B, M, 2)).astype(np.float32)
    return out, xyz.astype(np.float32), logits


def test_decode_matches_restatement():
    out, xyz, logits = _inputs(3, 500, 0)
    want = R.decode_and_reduce(out, xyz, logits)
    got = P.decode_proposals(torch.as_tensor(out).to(DEV), torch.as_tensor(xyz).to(DEV), torch.as_tensor(logits).to(DEV))
    for k in ("center", "angle", "size", "proposal_boxes", "proposal_scores", "boxes_bev"):
        g, w = got[k].cpu().numpy(), want[k]
        assert g.shape == w.shape, k
        assert np.abs(g - w).max() <= 2e-6 * max(1.0, np.abs(w).max()), k
    assert got["proposal_boxes"].shape == (3, 500, 8, 3)


def test_reduce_proposals_matches_restatement_and_pads_with_minus_one():
    out, xyz, logits = _inputs(2, 2048, 1)
    want = R.decode_and_reduce(out, xyz, logits)
    ep = P.decode_proposals(torch.as_tensor(out).to(DEV), torch.as_tensor(xyz).to(DEV), torch.as_tensor(logits).to(DEV))
    # NMS decisions on the restatement's own boxes / scores (bit-identical inputs), exact keep lists
    ep_w = dict(ep)
    ep_w["proposal_scores"] = torch.as_tensor(want["proposal_scores"]).to(DEV)
    ep_w["boxes_bev"] = torch.as_tensor(want["boxes_bev"]).to(DEV)
    got = P.reduce_proposals(ep_w)["nms_indices"].cpu().numpy()
    assert got.shape == (2, 2048) and got.dtype == np.int32
    assert np.array_equal(got, want["nms_indices"])
    n = (got >= 0).sum(1)
    assert (n <= 300).all() and (n > 10).all() and (got[:, 300:] == -1).all()
    # end to end on the kernel's own outputs: same keep set unless a pair sits within float noise of 0.5
    got2 = P.reduce_proposals(ep)["nms_indices"].cpu().numpy()
    assert (got2 == want["nms_indices"]).mean() > 0.99


def test_nms_tf_semantics_edge_cases():
    boxes = np.array([[[0, 0, 2, 2], [2, 2, 0, 0], [0, 0, 2, 2.0001], [5, 5, 5, 9], [0, 0, 1, 2], [10, 10, 12, 12]]], np.float32)
    scores = np.array([[0.9, 0.8, 0.7, 0.6, 0.5, 0.4]], np.float32)
    for thr in (0.5, 0.49999, 0.0):
        want = R.tf_nms(boxes[0], scores[0], 10, thr)
        keep, nk = P.non_max_suppression(torch.as_tensor(boxes).to(DEV), torch.as_tensor(scores).to(DEV), 10, thr)
        assert keep[0, :int(nk[0])].cpu().tolist() == want.tolist()
    # flipped corners are the same box, a zero-area box is kept and suppresses nothing,
    # IoU exactly 0.5 is NOT suppressed at threshold 0.5 (strict '>')
    keep, nk = P.non_max_suppression(torch.as_tensor(boxes).to(DEV), torch.as_tensor(scores).to(DEV), 10, 0.5)
    assert keep[0, :int(nk[0])].cpu().tolist() == [0, 3, 4, 5]
    keep, nk = P.non_max_suppression(torch.as_tensor(boxes).to(DEV), torch.as_tensor(scores).to(DEV), 2, 0.5)
    assert int(nk[0]) == 2


def test_bad_arguments():
    with pytest.raises(ValueError):
        P.decode_proposals(torch.zeros(1, 4, 50, device=DEV), torch.zeros(1, 4, 3, device=DEV))
    with pytest.raises(Exception):
        P.decode_proposals(torch.zeros(1, 4, 85), torch.zeros(1, 4, 3))


def _detections(F, n, seed):
    """clusters of near-duplicate detections, like the frustum stage's outputs on RPN proposals"""
    rng = np.random.default_rng(seed)
    det = np.zeros((F, n, 5))
    ncl = max(1, n // 6)
    for f in range(F):
        cx, cz = rng.uniform(-30, 30, ncl), rng.uniform(3, 60, ncl)
        ry = rng.uniform(-np.pi, np.pi, ncl)
        c = rng.integers(0, ncl, n)
        det[f, :, 0] = cx[c] + rng.normal(0, 0.4, n)
        det[f, :, 1] = cz[c] + rng.normal(0, 0.4, n)
        det[f, :, 2] = 3.9 * (1 + rng.normal(0, 0.1, n))
        det[f, :, 3] = 1.6 * (1 + rng.normal(0, 0.1, n))
        det[f, :, 4] = ry[c] + rng.normal(0, 0.2, n)
    return det, rng.standard_normal((F, n))


@pytest.mark.parametrize("F,n,thr", [(1, 1, 0.1), (3, 37, 0.1), (2, 200, 0.01), (4, 700, 0.1), (1, 64, 0.5)])
def test_nms_on_bev_matches_restatement(F, n, thr):
    from oracle import oracle as O
    det, sc = _detections(F, n, 100 + n)
    counts = np.array([n - (f * 5) % max(n, 1) if n > 20 else n for f in range(F)], np.int32)
    sc[0, : n // 2] = np.round(sc[0, : n // 2], 1)            # equal scores inside groups
    g, k, nk = P.nms_on_bev(torch.as_tensor(det).to(DEV), torch.as_tensor(sc).to(DEV), torch.as_tensor(counts), thr)
    g, k, nk = g.cpu().numpy(), k.cpu().numpy(), nk.cpu().numpy()
    for f in range(F):
        c = int(counts[f])
        wg, wk, margin = O.nms_on_bev(det[f, :c], sc[f, :c], thr, return_margin=True)
        assert margin > 1e-9, "fixture has a pair within float noise of the threshold"
        assert nk[f] == len(wk)
        assert np.array_equal(k[f, :nk[f]], wk) and (k[f, nk[f]:] == -1).all()
        assert np.array_equal(g[f, :c], wg) and (g[f, c:] == -1).all()
    assert (nk < counts).any() or n == 1                       # the fixture does merge something


def test_nms_on_bev_known_answers():
    # 0 and 2 overlap heavily, 1 is apart, 3 touches 0 only slightly (IoU 1/19 < 0.1), 4 == 1 rotated by pi
    det = np.array([[[0, 10, 4, 2, 0.0], [20, 10, 4, 2, 0.3], [0.5, 10, 4, 2, 0.0], [3.6, 10, 4, 2, 0.0],
                     [20, 10, 4, 2, 0.3 + np.pi]]])
    sc = np.array([[0.1, 0.5, 0.9, 0.2, 0.5]])
    g, k, nk = P.nms_on_bev(torch.as_tensor(det).to(DEV), torch.as_tensor(sc).to(DEV))
    assert g[0].cpu().tolist() == [0, 1, 0, 2, 1]
    assert int(nk[0]) == 3 and k[0].cpu().tolist() == [2, 1, 3, -1, -1]     # equal scores: the earlier one
    with pytest.raises(Exception):
        P.nms_on_bev(torch.as_tensor(det).to(DEV), torch.as_tensor(sc).to(DEV), None, 0.0)
