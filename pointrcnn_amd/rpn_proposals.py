"""RPN proposal post-processing on the GPU: the steps either side of the
rotated NMS that the reference runs as TensorFlow graph ops.

  decode_proposals   = BoxEncoder.tf_decode (utils/box_encoder.py:109-179) + rpn.py:108-113
  reduce_proposals   = RPN.reduce_proposals (models/rpn.py:209-241)
  nms_on_bev         = nms_on_bev / group_overlaps (test/test_frustum.py:191-218), the final
                       de-dup of the detections of a frame
Each is one C-ABI call (prcnn_rpn_decode_proposals, prcnn_nms_aabb_batched,
prcnn_bev_group_keep); there is no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib

NUM_HEADING_BIN = 12            # models/rpn.py:22-25
NUM_CENTER_BIN = 12
CENTER_SEARCH_RANGE = 3.0
HEADING_SEARCH_RANGE = float(np.pi)
TYPE_MEAN_SIZE = [[3.88311640418, 1.52563191462, 1.62856739989],      # utils/box_encoder.py:6-8 (l,h,w)
                  [0.84422524, 1.76255119, 0.66068622],
                  [1.76282397, 1.73698127, 0.59706367]]


def _f32(t, name, shape_tail=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.PointRCNNHipError("%s must be a GPU tensor (no CPU path exists)" % name)
    t = t.to(torch.float32).contiguous()
    if shape_tail is not None and tuple(t.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise ValueError("%s: expected trailing shape %s, got %s" % (name, shape_tail, tuple(t.shape)))
    return t


def decode_proposals(output, fg_points_xyz, fg_logits=None, num_center_bin=NUM_CENTER_BIN,
                     num_heading_bin=NUM_HEADING_BIN, mean_sizes=TYPE_MEAN_SIZE,
                     center_search_range=CENTER_SEARCH_RANGE, heading_search_range=HEADING_SEARCH_RANGE):
    """output (B,M,4*NCB+1+2*NHB+4*NSC) head activations (layout of models/rpn.py:76-104),
    fg_points_xyz (B,M,3), fg_logits (B,M,2) or None -> dict with center (B,M,3), angle (B,M),
    size (B,M,3), proposal_boxes (B,M,8,3), proposal_scores (B,M), boxes_bev (B,M,4)."""
    nsc = len(mean_sizes)
    row = 4 * num_center_bin + 1 + 2 * num_heading_bin + 4 * nsc
    output = _f32(output, "head output", (row,))
    B, M = output.shape[0], output.shape[1]
    xyz = _f32(fg_points_xyz, "fg_points_xyz", (M, 3))
    logits = None if fg_logits is None else _f32(fg_logits, "fg_logits", (M, 2))
    dev = output.device
    mk = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    center, angle, size = mk(B, M, 3), mk(B, M), mk(B, M, 3)
    corners, conf, bev = mk(B, M, 8, 3), mk(B, M), mk(B, M, 4)
    ms = (ctypes.c_float * (nsc * 3))(*[float(v) for r in mean_sizes for v in r])
    p = lambda t: None if t is None else t.data_ptr()
    _lib.check(_lib.lib().prcnn_rpn_decode_proposals(
        B, M, num_center_bin, num_heading_bin, nsc, float(center_search_range), float(heading_search_range), ms,
        p(output), p(xyz), p(logits), p(center), p(angle), p(size), p(corners), p(conf), p(bev),
        torch.cuda.current_stream().cuda_stream), "rpn_decode_proposals")
    return {"center": center, "angle": angle, "size": size, "proposal_boxes": corners,
            "proposal_scores": conf, "boxes_bev": bev}


def non_max_suppression(boxes, scores, max_output_size, iou_threshold=0.5):
    """tf.image.non_max_suppression per frame: boxes (B,M,4) f32, scores (B,M) ->
    keep (B,max_output_size) int64 and n_keep (B,) int32."""
    boxes = _f32(boxes, "boxes", (4,))
    if boxes.dim() != 3:
        raise ValueError("boxes must be (B,M,4)")
    B, M, _ = boxes.shape
    sc = scores.to(torch.float32).to(torch.float64).contiguous()          # exact widening
    if tuple(sc.shape) != (B, M) or not sc.is_cuda:
        raise ValueError("scores must be a (B,M) GPU tensor")
    L = _lib.lib()
    max_output_size = int(max_output_size)
    keep = torch.empty((B, max(max_output_size, 1)), dtype=torch.int64, device=boxes.device)
    nk = torch.empty(B, dtype=torch.int32, device=boxes.device)
    wsb = L.prcnn_nms_rotate_batched_workspace_bytes(B, M)
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device=boxes.device)
    off = (-ws.data_ptr()) % 256
    _lib.check(L.prcnn_nms_aabb_batched(B, M, boxes.data_ptr(), sc.data_ptr(), float(iou_threshold), max_output_size,
                                        keep.data_ptr(), nk.data_ptr(), ws.data_ptr() + off, wsb,
                                        torch.cuda.current_stream().cuda_stream), "nms_aabb_batched")
    return keep, nk


def reduce_proposals(end_points, max_output_size=300, iou_threshold=0.5):
    """models/rpn.py:209-241: adds 'nms_indices' (B,M) int32, the kept proposal indices in
    selection order padded with -1 (tf.pad constant_values=-1), to the dict of decode_proposals."""
    conf, bev = end_points["proposal_scores"], end_points["boxes_bev"]
    B, M = conf.shape
    keep, nk = non_max_suppression(bev, conf, max_output_size, iou_threshold)
    K = keep.shape[1]
    valid = torch.arange(K, device=conf.device).unsqueeze(0) < nk.view(B, 1)
    idx = torch.full((B, M), -1, dtype=torch.int32, device=conf.device)
    idx[:, :min(K, M)] = torch.where(valid, keep, torch.full_like(keep, -1))[:, :M].to(torch.int32)
    out = dict(end_points)
    out["nms_indices"] = idx
    return out


def nms_on_bev(boxes, scores, counts=None, iou_threshold=0.1):
    """test/test_frustum.py:206-218 for a batch of frames.  boxes (F,n,5) rows [x, z, l, w, ry]
    (obj.t[0], obj.t[2], obj.l, obj.w, obj.ry), scores (F,n), counts (F,) detections present per
    frame or None -> (group_id (F,n) int32, keep (F,n) int32 padded with -1, n_keep (F,) int32):
    keep[f, g] is the surviving detection of group g, in the reference's output order."""
    if not isinstance(boxes, torch.Tensor) or not boxes.is_cuda:
        raise _lib.PointRCNNHipError("boxes must be a GPU tensor (no CPU path exists)")
    if boxes.dim() != 3 or boxes.shape[-1] != 5:
        raise ValueError("boxes must be (F,n,5)")
    F, n, _ = boxes.shape
    bx = boxes.to(torch.float64).contiguous()
    sc = scores.to(torch.float64).contiguous()
    if tuple(sc.shape) != (F, n) or not sc.is_cuda:
        raise ValueError("scores must be a (F,n) GPU tensor")
    cp = None
    if counts is not None:
        counts = counts.to(device=bx.device, dtype=torch.int32).contiguous()
        if tuple(counts.shape) != (F,):
            raise ValueError("counts must be (F,)")
        cp = counts.data_ptr()
    L = _lib.lib()
    group = torch.empty((F, n), dtype=torch.int32, device=bx.device)
    keep = torch.empty((F, n), dtype=torch.int32, device=bx.device)
    nk = torch.empty(F, dtype=torch.int32, device=bx.device)
    wsb = L.prcnn_nms_rotate_batched_workspace_bytes(F, n)
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device=bx.device)
    off = (-ws.data_ptr()) % 256
    _lib.check(L.prcnn_bev_group_keep(F, n, cp, bx.data_ptr(), sc.data_ptr(), float(iou_threshold), group.data_ptr(),
                                      keep.data_ptr(), nk.data_ptr(), ws.data_ptr() + off, wsb,
                                      torch.cuda.current_stream().cuda_stream), "bev_group_keep")
    return group, keep, nk
