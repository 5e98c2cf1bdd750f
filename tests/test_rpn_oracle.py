"""CPU known-answer tests of oracle/rpn_numpy.py (the checker of tests/test_rpn_proposals_gpu.py).
Parity unpinned (TensorFlow absent, no reference fixture): the expected values below are worked by
hand from utils/box_encoder.py:109-179, models/model_util.py:55-82 and TF's NonMaxSuppression."""
import numpy as np

from oracle import rpn_numpy as R


def test_decode_known_answers():
    out = np.zeros((1, 2, 85), np.float32)
    # proposal 1: x bin 7 residual 0.5, z bin 11 residual -0.5, y residual 0.3, heading bin 6
    # residual 0.25, size cluster 2 residuals (0.1, 0, -0.5)
    out[0, 1, 7] = 5; out[0, 1, 12 + 7] = 0.5
    out[0, 1, 24 + 11] = 5; out[0, 1, 36 + 11] = -0.5
    out[0, 1, 48] = 0.3
    out[0, 1, 49 + 6] = 5; out[0, 1, 61 + 6] = 0.25
    out[0, 1, 73 + 2] = 5; out[0, 1, 76 + 6:76 + 9] = (0.1, 0.0, -0.5)
    c, a, s = R.tf_decode(R.split_head(out))
    np.testing.assert_allclose(c[0, 0], [-2.75, 0.0, -2.75], atol=1e-6)      # bin 0: 0.25 - 3
    np.testing.assert_allclose(a[0, 0], -np.pi, atol=1e-6)
    np.testing.assert_allclose(s[0, 0], R.TYPE_MEAN_SIZE[0], atol=1e-6)
    np.testing.assert_allclose(c[0, 1], [7 * 0.5 + 0.25 - 3 + 0.25, 0.3, 11 * 0.5 + 0.25 - 3 - 0.25], atol=1e-6)
    np.testing.assert_allclose(a[0, 1], 6.25 * (np.pi / 6) - np.pi, atol=1e-6)
    np.testing.assert_allclose(s[0, 1], R.TYPE_MEAN_SIZE[2] * np.array([1.1, 1.0, 0.5]), atol=1e-6)


def test_corners_known_answers():
    c = np.array([[1.0, 2.0, 3.0]], np.float32)
    s = np.array([[4.0, 2.0, 1.0]], np.float32)
    k0 = R.get_box3d_corners(c, np.array([0.0], np.float32), s)[0]
    np.testing.assert_allclose(k0[0], [3.0, 2.0, 3.5], atol=1e-6)
    np.testing.assert_allclose(k0[6], [-1.0, 0.0, 2.5], atol=1e-6)
    k90 = R.get_box3d_corners(c, np.array([np.pi / 2], np.float32), s)[0]     # x' = z, z' = -x
    np.testing.assert_allclose(k90[0], [1.5, 2.0, 1.0], atol=1e-5)
    assert np.allclose(k0.mean(0), [1.0, 1.0, 3.0], atol=1e-6)


def test_tf_nms_known_answers():
    boxes = np.array([[0, 0, 2, 2], [2, 2, 0, 0], [0, 0, 2, 2.0001], [5, 5, 5, 9], [0, 0, 1, 2], [10, 10, 12, 12]], np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.6, 0.5, 0.4], np.float32)
    assert R.tf_nms(boxes, scores, 10, 0.5).tolist() == [0, 3, 4, 5]          # IoU 0.5 is not > 0.5
    assert R.tf_nms(boxes, scores, 10, 0.4).tolist() == [0, 3, 5]
    assert R.tf_nms(boxes, scores, 2, 0.5).tolist() == [0, 3]
    assert R.tf_nms(boxes, np.full(6, 0.5, np.float32), 10, 0.5).tolist() == [0, 3, 4, 5]   # ties: lower index


def test_reduce_pads_with_minus_one():
    rng = np.random.default_rng(0)
    out = rng.standard_normal((2, 64, 85)).astype(np.float32)
    xyz = rng.uniform(-1, 1, (2, 64, 3)).astype(np.float32)
    ep = R.decode_and_reduce(out, xyz, rng.standard_normal((2, 64, 2)).astype(np.float32), max_out=8)
    idx = ep['nms_indices']
    assert idx.shape == (2, 64) and (idx[:, 8:] == -1).all()
    for b in range(2):
        k = idx[b][idx[b] >= 0]
        assert len(set(k.tolist())) == len(k) and (np.diff(ep['proposal_scores'][b][k]) <= 0).all()
    assert (ep['proposal_scores'] > 0).all() and (ep['proposal_scores'] <= 1).all()


def test_nms_on_bev_restatement_known_answers():
    from oracle import oracle as O
    a = [0, 10, 4, 2, 0.0]
    assert abs(O.bev_iou(a, a) - 1.0) < 1e-12
    assert abs(O.bev_iou(a, [2, 10, 4, 2, 0.0]) - 1.0 / 3.0) < 1e-12          # half overlap: 4/(8+8-4)
    assert abs(O.bev_iou(a, [0, 10, 4, 2, np.pi]) - 1.0) < 1e-12
    assert abs(O.bev_iou([0, 0, 2, 2, 0.0], [0, 0, 2, 2, np.pi / 4]) - (8 * (np.sqrt(2) - 1)) / (8 - 8 * (np.sqrt(2) - 1))) < 1e-12
    assert O.bev_iou(a, [10, 10, 4, 2, 0.0]) == 0.0
    det = np.array([a, [20, 10, 4, 2, 0.3], [0.5, 10, 4, 2, 0.0], [3.6, 10, 4, 2, 0.0], [20, 10, 4, 2, 0.3 + np.pi]])
    g, k = O.nms_on_bev(det, [0.1, 0.5, 0.9, 0.2, 0.5])
    assert g.tolist() == [0, 1, 0, 2, 1] and k.tolist() == [2, 1, 3]
    g, k = O.nms_on_bev(np.zeros((0, 5)), np.zeros(0))
    assert len(g) == 0 and len(k) == 0
