"""GPU parity: rotated NMS through the C ABI vs the float64 oracle
(identical keep list except for pairs within 1e-5 of the IoU threshold)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from pointrcnn_amd import synthetic as S
from pointrcnn_amd.nms_rotate import nms_rotate, nms_rotate_cpu

pytestmark = pytest.mark.gpu


def _check(boxes, scores, thr, mx):
    got = nms_rotate_cpu(boxes, scores, thr, mx)
    want, margin = O.nms_rotate_cpu(boxes, scores, thr, mx, return_margin=True)
    assert got.dtype == np.int64
    if not np.array_equal(got, want):
        assert margin < 1e-5, "keep lists differ although no pair is within 1e-5 of the threshold"
    return got, want


def test_reference_demo(cuda):
    boxes = np.array([[50, 40, 100, 100, 0], [60, 50, 100, 100, 0],
                      [50, 30, 100, 100, -45.], [200, 190, 100, 100, 0.]])
    scores = np.array([0.99, 0.88, 0.66, 0.77])
    assert list(nms_rotate_cpu(boxes, scores, 0.7, 1)) == [0]
    assert list(nms_rotate_cpu(boxes, scores, 0.7, 10)) == [0, 1, 3, 2]
    assert list(nms_rotate_cpu(boxes, scores, 0.68, 10)) == [0, 3, 2]
    k = nms_rotate(torch.tensor(boxes, device=cuda), torch.tensor(scores, device=cuda), 0.7, 1)
    assert k.dtype == torch.int64 and k.tolist() == [0]


def test_edge_cases(cuda):
    b = np.array([[0, 0, 5, 20, 0.], [0, 0, 5, 20, 0.], [0, 0, 5, 20, 180.], [0, 0, 20, 5, 90.],
                  [0, 0, 1, 1, 10.], [100, 0, 5, 20, 0.], [102.5, 0, 5, 20, 0.], [0, 20, 5, 20, 0.]])
    s = np.array([.9, .8, .7, .6, .5, .4, .3, .2])
    for thr in (0.99, 0.5, 0.3, 0.009, 1e-9):
        for mx in (1, 3, 100):
            g, w = _check(b, s, thr, mx)
            assert np.array_equal(g, w)
    # identical / wrapped boxes are suppressed, the contained small box and the touching box survive
    assert list(nms_rotate_cpu(b, s, 0.5, 100)) == [0, 4, 5, 6, 7]
    assert len(nms_rotate_cpu(b[:0], s[:0], 0.5, 10)) == 0
    assert len(nms_rotate_cpu(b, s, 0.5, 0)) == 0
    assert list(nms_rotate_cpu(b, s, 0.0, 100)) == [0]         # iou >= 0 always: everything after the first goes


@pytest.mark.parametrize("N,thr,mx", [(2048, 0.85, 300), (2048, 0.8, 100), (2048, 0.7, 10), (2048, 0.7, 300),
                                      (777, 0.3, 1000), (130, 0.5, 64),
                                      # around the 512-rank head of the two-stage evaluation: the head ends the
                                      # sweep (max_out reached at / just before its last rank), just fails to, or
                                      # holds every box
                                      (513, 0.95, 512), (1024, 0.99, 511), (1024, 0.99, 512), (1024, 0.99, 513),
                                      (512, 0.5, 512), (511, 0.9, 1000), (3000, 0.6, 400)])
def test_reference_scale(cuda, N, thr, mx):
    boxes, scores = S.nms_boxes(seed=N, N=N, clusters=max(2, N // 60))
    got, want = _check(boxes, scores, thr, mx)
    assert np.array_equal(got, want)


def test_batched_frames_finish_in_head_or_tail_independently(cuda):
    """prcnn_nms_rotate_batched on frames of very different density: sparse frames end inside the
    512-rank head (their done flag silences the tail launches), dense ones (2 clusters: almost
    everything suppressed) need the full mask -- every frame equals its own single-frame oracle."""
    from pointrcnn_amd.nms_rotate import nms_rotate_batched
    N = 1500
    frames = [S.nms_boxes(seed=50 + i, N=N, clusters=c) for i, c in enumerate((300, 2, 40, 1, 700, 3))]
    boxes = torch.as_tensor(np.stack([f[0] for f in frames])).to(cuda)
    scores = torch.as_tensor(np.stack([f[1] for f in frames])).to(cuda)
    for thr, mx in ((0.8, 100), (0.5, 300), (0.3, 1500)):
        keep, nk = nms_rotate_batched(boxes, scores, thr, mx)
        keep, nk = keep.cpu().numpy(), nk.cpu().numpy()
        for i, (b, s) in enumerate(frames):
            want = O.nms_rotate_cpu(b, s, thr, mx)
            assert nk[i] == len(want) and np.array_equal(keep[i, :nk[i]], want), (thr, mx, i)


def test_headline_9000_to_100(cuda):
    boxes, scores = S.nms_boxes(seed=9000, N=9000)
    got, want = _check(boxes, scores, 0.8, 100)
    assert np.array_equal(got, want) and len(got) == 100
    got, want = _check(boxes, scores, 0.3, 9000)               # deep sweep: most boxes suppressed
    assert np.array_equal(got, want)
    # idempotence: NMS of the kept set keeps everything
    again = nms_rotate_cpu(boxes[got], scores[got], 0.3, 9000)
    assert np.array_equal(again, np.arange(len(got)))


def test_score_ties_follow_reversed_stable_argsort(cuda):
    """Equal scores (saturated / duplicated confidences): the library orders them higher index first,
    i.e. scores.argsort(kind='stable')[::-1] -- the reference's line (nms_rotate.py:41) with the
    stable sort; checked against the oracle's DEFAULT order and against NumPy on a short array."""
    boxes, _ = S.nms_boxes(seed=3, N=200, clusters=4)
    scores = np.repeat(np.linspace(1, 0, 50), 4)
    want = O.nms_rotate_cpu(boxes, scores, 0.5, 100)
    got = nms_rotate_cpu(boxes, scores, 0.5, 100)
    assert np.array_equal(got, want)
    assert got[0] == 3                                          # of the four boxes with score 1.0 the last goes first
    far = np.array([[100.0 * i, 0, 2, 1, 0.] for i in range(12)])                 # disjoint: everything is kept
    s12 = np.array([.5, .9, .5, .5, .9, .1, .5, .9, .1, .5, .5, .9])
    assert np.array_equal(nms_rotate_cpu(far, s12, 0.5, 100), s12.argsort(kind='stable')[::-1])
    assert np.array_equal(nms_rotate_cpu(far, s12, 0.5, 100), O.nms_rotate_cpu(far, s12, 0.5, 100))


def test_pack_detections_kernel_equals_host_composition(cuda):
    """prcnn_pack_detections (one launch) against dist.pack_detections' torch composition on
    host tensors, which the world-size-2 gloo test exercises."""
    from pointrcnn_amd import dist as pdist
    g = torch.Generator().manual_seed(3)
    S_, N = 5, 300
    b3 = torch.randn(S_, N, 7, generator=g, dtype=torch.float64)
    sc = torch.rand(S_, N, generator=g, dtype=torch.float64)
    keep = torch.randint(0, N, (S_, pdist.MAX_DET), generator=g)
    nk = torch.tensor([0, 1, 37, pdist.MAX_DET, 99], dtype=torch.int32)
    ids = torch.arange(10, 10 + S_, dtype=torch.float32)
    want = pdist.pack_detections(ids, b3, sc, keep, nk)
    got = pdist.pack_detections(ids.to(cuda), b3.to(cuda), sc.to(cuda), keep.to(cuda), nk.to(cuda))
    assert got.is_cuda and torch.equal(got.cpu(), want)
