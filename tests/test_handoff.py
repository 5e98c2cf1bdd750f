"""Formats of SURVEY.md 8f rank 4 (host side): record keys of test/test_rpn.py:184-195 and the
KITTI label line of test/test_frustum.py:160-163 (expected strings worked by hand)."""
import os

import numpy as np

from pointrcnn_amd import handoff as H


def test_kitti_label_line_and_files(tmp_path):
    line = H.kitti_label_line("Car", [1, 2, 3.5, 4], 1.5, 1.6, 3.9, (1.0, 1.65, 20.0), -0.1, 0.875)
    assert line == ("Car -1 -1 -10 1.000000 2.000000 3.500000 4.000000 "
                    "1.500000 1.600000 3.900000 1.000000 1.650000 20.000000 -0.100000 0.875000")
    objs = {7: [dict(type_label="Car", box_2d=[1, 2, 3.5, 4], h=1.5, w=1.6, l=3.9, t=(1.0, 1.65, 20.0), ry=-0.1, score=0.875)] * 2,
            12: []}
    paths = H.write_detection_results(str(tmp_path), objs)
    assert sorted(os.path.basename(p) for p in paths) == ["000007.txt", "000012.txt"]
    assert open(os.path.join(str(tmp_path), "data", "000007.txt")).read() == (line + "\n") * 2
    assert open(os.path.join(str(tmp_path), "data", "000012.txt")).read() == ""
    H.fill_files(os.path.join(str(tmp_path), "data"), ["000007.txt", "000099.txt"])
    assert os.path.getsize(os.path.join(str(tmp_path), "data", "000007.txt")) > 0
    assert os.path.exists(os.path.join(str(tmp_path), "data", "000099.txt"))
    assert H.write_detection_results(None, objs) == []


def test_frame_records_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    B, N, M = 2, 64, 16
    logits = rng.standard_normal((B, N, 2)).astype(np.float32)
    ep = {"center": rng.standard_normal((B, M, 3)).astype(np.float32), "angle": rng.standard_normal((B, M)).astype(np.float32),
          "size": rng.random((B, M, 3)).astype(np.float32), "proposal_boxes": rng.standard_normal((B, M, 8, 3)).astype(np.float32),
          "nms_indices": np.full((B, M), -1, np.int32), "proposal_scores": rng.random((B, M)).astype(np.float32)}
    recs = H.frame_records(["000001", "000002"], logits, rng.integers(0, N, (B, M)), ep, rng.integers(0, 20000, (B, N)))
    assert [tuple(r.keys()) for r in recs] == [H.RECORD_KEYS] * 2
    assert np.array_equal(recs[1]["segmentation"], logits[1].argmax(1)) and recs[0]["proposal_boxes"].shape == (M, 8, 3)
    path = H.save_frame_record(str(tmp_path), recs[0])
    assert os.path.basename(path) == "000001.pkl"
    back = H.load_frame_record(path)
    for k in H.RECORD_KEYS[1:]:
        assert np.array_equal(back[k], recs[0][k])
