"""Data formats either side of the hot path: the per-frame RPN -> RCNN
hand-off record of the reference (test/test_rpn.py:184-200) and the KITTI label writer
(test/test_frustum.py:153-175).  Host-side Python like the reference's; the tensors come from
`rpn_proposals.decode_proposals` / `reduce_proposals` (GPU) and are copied to the host once.
"""
import os
import pickle

import numpy as np

RECORD_KEYS = ("frame_id", "segmentation", "fg_indices", "centers", "angles", "sizes",
               "proposal_boxes", "nms_indices", "scores", "pc_choices")


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def frame_records(frame_ids, foreground_logits, fg_point_indices, end_points, pc_choices):
    """One dict per frame with the reference's keys (test_rpn.py:184-195).
    foreground_logits (B,N,2); fg_point_indices (B,M); end_points = output of
    rpn_proposals.reduce_proposals; pc_choices (B,N) the dataset's point sub-sampling."""
    seg = np.argmax(_np(foreground_logits), 2)                   # test_rpn.py:172
    idx = _np(fg_point_indices)
    cen, ang, siz = _np(end_points["center"]), _np(end_points["angle"]), _np(end_points["size"])
    cor, nms, sco = _np(end_points["proposal_boxes"]), _np(end_points["nms_indices"]), _np(end_points["proposal_scores"])
    pcc = _np(pc_choices)
    return [{"frame_id": frame_ids[i], "segmentation": seg[i], "fg_indices": idx[i], "centers": cen[i],
             "angles": ang[i], "sizes": siz[i], "proposal_boxes": cor[i], "nms_indices": nms[i],
             "scores": sco[i], "pc_choices": pcc[i]} for i in range(len(frame_ids))]


def save_frame_record(directory, record, protocol=2):
    """'<frame_id>.pkl' as test_rpn.py:198-199 writes it (protocol 2: readable by the reference's
    Python 2)."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, str(record["frame_id"]) + ".pkl")
    with open(path, "wb") as f:
        pickle.dump({k: record[k] for k in RECORD_KEYS}, f, protocol=protocol)
    return path


def load_frame_record(path):
    """Reads a record written by save_frame_record (files of this package only)."""
    with open(path, "rb") as f:
        rec = pickle.load(f)
    missing = [k for k in RECORD_KEYS if k not in rec]
    if missing:
        raise ValueError("%s: missing keys %s" % (path, missing))
    return rec


def kitti_label_line(type_label, box2d, h, w, l, t, ry, score):
    """test_frustum.py:160-163."""
    s = type_label + " -1 -1 -10 "
    s += "%f %f %f %f " % (box2d[0], box2d[1], box2d[2], box2d[3])
    s += "%f %f %f %f %f %f %f %f" % (h, w, l, t[0], t[1], t[2], ry, score)
    return s


def write_detection_results(result_dir, detection_objects):
    """test_frustum.py:153-175: detection_objects maps frame idx -> list of objects with
    type_label, box_2d, h, w, l, t, ry, score (attributes or dict keys); writes
    <result_dir>/data/%06d.txt and returns the paths."""
    if result_dir is None:
        return []
    get = lambda o, k: o[k] if isinstance(o, dict) else getattr(o, k)
    out_dir = os.path.join(result_dir, "data")
    os.makedirs(out_dir, exist_ok=True)
    paths = []
    for idx, objs in detection_objects.items():
        path = os.path.join(out_dir, "%06d.txt" % int(idx))
        with open(path, "w") as f:
            for o in objs:
                f.write(kitti_label_line(get(o, "type_label"), get(o, "box_2d"), get(o, "h"), get(o, "w"), get(o, "l"),
                                         get(o, "t"), get(o, "ry"), get(o, "score")) + "\n")
        paths.append(path)
    return paths


def fill_files(output_dir, filenames):
    """test_frustum.py:177-183: empty label files for frames without detections."""
    for name in filenames:
        path = os.path.join(output_dir, name)
        if not os.path.exists(path):
            open(path, "w").close()
