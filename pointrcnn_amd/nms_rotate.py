"""Rotated BEV NMS with the reference's call surface (utils/nms_rotate.py).

  nms_rotate(decode_boxes, scores, iou_threshold, max_output_size, ...)   :10-37
  nms_rotate_cpu(boxes, scores, iou_threshold, max_output_size)           :39-73
Both names are kept so the reference's callers (dataset/frustum_dataset.py:129,
dataset/rcnn_dataset.py:194, train/stat_proposal.py:47) work unchanged; both run the
gfx950 kernels of libpointrcnn_hip.so (there is no CPU implementation here).  NumPy in
-> NumPy int64 out (as the reference), torch GPU tensors in -> torch int64 GPU tensor out.
"""
import numpy as np
import torch

from . import _lib


def _run(boxes, scores, iou_threshold, max_output_size, pad=False):
    n = boxes.shape[0]
    L = _lib.lib()
    dev = boxes.device
    max_output_size = int(max_output_size)
    cap = max(max_output_size if pad else min(n, max_output_size), 0)
    keep = torch.empty(max(cap, 1), dtype=torch.int64, device=dev)
    nk = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = L.prcnn_nms_rotate_workspace_bytes(n)
    ws = torch.empty(wsb + 256, dtype=torch.uint8, device=dev)
    off = (-ws.data_ptr()) % 256
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.prcnn_nms_rotate(n, boxes.data_ptr(), scores.data_ptr(), float(iou_threshold),
                                  max_output_size, keep.data_ptr(), nk.data_ptr(),
                                  ws.data_ptr() + off, wsb, st), "nms_rotate")
    return keep, nk


def nms_rotate_batched(boxes, scores, iou_threshold, max_output_size, workspace=None):
    """One NMS per frame for a batch of frames in one set of launches, no host sync.
    boxes (B,N,5) / scores (B,N) float64 GPU tensors -> keep (B,max_output_size) int64,
    n_keep (B,) int32.  `workspace` (uint8 GPU tensor) may be passed to avoid allocation."""
    if not (boxes.is_cuda and boxes.dtype == torch.float64 and boxes.is_contiguous() and boxes.dim() == 3
            and boxes.shape[2] == 5 and scores.dtype == torch.float64 and scores.is_contiguous()
            and scores.shape == boxes.shape[:2]):
        raise ValueError("nms_rotate_batched expects contiguous float64 GPU tensors (B,N,5) and (B,N)")
    B, n, _ = boxes.shape
    L = _lib.lib()
    max_output_size = int(max_output_size)
    keep = torch.empty((B, max(max_output_size, 1)), dtype=torch.int64, device=boxes.device)
    nk = torch.empty(B, dtype=torch.int32, device=boxes.device)
    wsb = L.prcnn_nms_rotate_batched_workspace_bytes(B, n)
    if workspace is None or workspace.numel() < wsb + 256:
        workspace = torch.empty(wsb + 256, dtype=torch.uint8, device=boxes.device)
    off = (-workspace.data_ptr()) % 256
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.prcnn_nms_rotate_batched(B, n, boxes.data_ptr(), scores.data_ptr(), float(iou_threshold),
                                          max_output_size, keep.data_ptr(), nk.data_ptr(),
                                          workspace.data_ptr() + off, wsb, st), "nms_rotate_batched")
    return keep, nk


def nms_rotate_padded(boxes, scores, iou_threshold, max_output_size):
    """Asynchronous form for pipelines: float64 GPU tensors in, (keep (max_output_size,)
    int64, n_keep (1,) int32) GPU tensors out, no host synchronisation.  Entries of keep
    beyond n_keep are unspecified."""
    if not (boxes.is_cuda and boxes.dtype == torch.float64 and boxes.is_contiguous()
            and scores.dtype == torch.float64 and scores.is_contiguous()):
        raise ValueError("nms_rotate_padded expects contiguous float64 GPU tensors")
    return _run(boxes, scores, iou_threshold, max_output_size, pad=True)


def nms_rotate_cpu(boxes, scores, iou_threshold, max_output_size):
    """Same signature and result as the reference's nms_rotate_cpu (name kept for drop-in
    use; the work runs on the GPU).  boxes (N,5) [x_c, y_c, w, h, theta_deg], scores (N,)."""
    is_np = not isinstance(boxes, torch.Tensor)
    if is_np:
        if not torch.cuda.is_available():
            raise _lib.PointRCNNHipError("nms_rotate needs a GPU (no CPU path exists)")
        b = torch.as_tensor(np.ascontiguousarray(boxes, np.float64)).cuda()
        s = torch.as_tensor(np.ascontiguousarray(scores, np.float64)).cuda()
    else:
        if not boxes.is_cuda:
            raise _lib.PointRCNNHipError("nms_rotate needs GPU tensors (no CPU path exists)")
        b = boxes.to(torch.float64).contiguous()
        s = scores.to(device=b.device, dtype=torch.float64).contiguous()
    if b.dim() != 2 or b.shape[1] != 5 or s.dim() != 1 or s.shape[0] != b.shape[0]:
        raise ValueError("nms_rotate expects boxes (N,5) and scores (N,)")
    keep, nk = _run(b, s, iou_threshold, max_output_size)
    k = int(nk.item())
    keep = keep[:k]
    return keep.cpu().numpy().astype(np.int64) if is_np else keep


def nms_rotate(decode_boxes, scores, iou_threshold, max_output_size,
               use_angle_condition=False, angle_threshold=0, use_gpu=False, gpu_id=0):
    """utils/nms_rotate.py:10-37.  The reference's use_gpu=True branch calls an undefined
    `rotate_gpu_nms` (:81,:89) and cannot run; here both settings take the same device
    path and the angle-condition arguments are accepted and ignored, as on its CPU branch."""
    return nms_rotate_cpu(decode_boxes, scores, iou_threshold, max_output_size)
