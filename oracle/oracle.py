"""ctypes front-end of the CPU oracle (oracle/pointrcnn_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.

Every wrapper takes / returns NumPy arrays with the reference op's shapes
(see include/pointrcnn_hip.h for the reference file:line of each op).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when the reference tree is mounted)."""
    src = os.path.join(_HERE, "pointrcnn_oracle.c")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.oracle_rotated_intersection.restype = ctypes.c_double
        _lib.oracle_rotated_iou.restype = ctypes.c_double
        _lib.oracle_nms_rotate.restype = ctypes.c_int
        _lib.oracle_bev_iou.restype = ctypes.c_double
        _lib.oracle_bev_group_keep.restype = ctypes.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def farthest_point_sample(npoint, inp):
    inp = _f32(inp)
    b, n, _ = inp.shape
    out = np.zeros((b, npoint), np.int32)
    temp = np.empty(n, np.float32)
    lib().oracle_farthest_point_sample(b, n, npoint, _p(inp), _p(temp), _p(out))
    return out


def gather_point(inp, idx):
    inp, idx = _f32(inp), _i32(idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = np.empty((b, m, 3), np.float32)
    lib().oracle_gather_point(b, n, m, _p(inp), _p(idx), _p(out))
    return out


def query_ball_point(radius, nsample, xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    cnt = np.zeros((b, m), np.int32)
    lib().oracle_query_ball_point(b, n, m, ctypes.c_float(radius), nsample,
                                  _p(xyz1), _p(xyz2), _p(idx), _p(cnt))
    return idx, cnt


def group_point(points, idx):
    points, idx = _f32(points), _i32(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    lib().oracle_group_point(b, n, c, m, ns, _p(points), _p(idx), _p(out))
    return out


def three_nn(xyz1, xyz2):
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    lib().oracle_three_nn(b, n, m, _p(xyz1), _p(xyz2), _p(dist), _p(idx))
    return dist, idx


def fp_weights(dist):
    dist = _f32(dist)
    w = np.empty_like(dist)
    lib().oracle_fp_weights(int(dist.size // 3), _p(dist), _p(w))
    return w


def three_interpolate(points, idx, weight):
    points, idx, weight = _f32(points), _i32(idx), _f32(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    lib().oracle_three_interpolate(b, m, c, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


def select_top_k(k, dist):
    """tf_grouping.py:23-32 -> (idx (b,m,n) int32, dist_out (b,m,n) f32)."""
    dist = _f32(dist)
    b, m, n = dist.shape
    idx = np.empty((b, m, n), np.int32)
    val = np.empty((b, m, n), np.float32)
    lib().oracle_selection_sort(b, n, m, int(k), _p(dist), _p(idx), _p(val))
    return idx, val


def knn_point(k, xyz1, xyz2):
    """tf_grouping.py:49-74 -> (val (b,m,k) f32, idx (b,m,k) int32)."""
    xyz1, xyz2 = _f32(xyz1), _f32(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    val = np.empty((b, m, int(k)), np.float32)
    idx = np.empty((b, m, int(k)), np.int32)
    lib().oracle_knn_point(b, n, m, int(k), _p(xyz1), _p(xyz2), _p(val), _p(idx))
    return val, idx


def gather_point_grad(inp, idx, out_g):
    """Argument order of the reference op GatherPointGrad (tf_sampling.py:44-48): only
    inp's shape is used."""
    idx, out_g = _i32(idx), _f32(out_g)
    b, n = inp.shape[0], inp.shape[1]
    m = idx.shape[1]
    g = np.empty((b, n, 3), np.float32)
    lib().oracle_gather_point_grad(b, n, m, _p(out_g), _p(idx), _p(g))
    return g


def group_point_grad(points, idx, grad_out):
    """GroupPointGrad (tf_grouping.py:43-47): only points' shape is used."""
    idx, grad_out = _i32(idx), _f32(grad_out)
    b, n, c = points.shape
    _, m, ns = idx.shape
    g = np.empty((b, n, c), np.float32)
    lib().oracle_group_point_grad(b, n, c, m, ns, _p(grad_out), _p(idx), _p(g))
    return g


def three_interpolate_grad(points, idx, weight, grad_out):
    """ThreeInterpolateGrad (tf_interpolate.py:30-35): only points' shape is used."""
    idx, weight, grad_out = _i32(idx), _f32(weight), _f32(grad_out)
    b, m, c = points.shape
    n = idx.shape[1]
    g = np.empty((b, m, c), np.float32)
    lib().oracle_three_interpolate_grad(b, n, c, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


def rotated_intersection(b1, b2):
    b1 = np.ascontiguousarray(b1, np.float64)
    b2 = np.ascontiguousarray(b2, np.float64)
    return lib().oracle_rotated_intersection(_p(b1), _p(b2))


def rotated_iou(b1, b2):
    b1 = np.ascontiguousarray(b1, np.float64)
    b2 = np.ascontiguousarray(b2, np.float64)
    return lib().oracle_rotated_iou(_p(b1), _p(b2))


def nms_rotate_cpu(boxes, scores, iou_threshold, max_output_size, order=None,
                   return_margin=False):
    """utils/nms_rotate.py:39-73.  order defaults to scores.argsort()[::-1] (:41), with the STABLE
    sort: NumPy's default (introsort) leaves the order of equal scores unspecified, the stable
    variant puts them higher index first after the reversal -- the convention the library follows."""
    boxes = np.ascontiguousarray(boxes, np.float64)
    scores = np.asarray(scores)
    if order is None:
        order = scores.argsort(kind='stable')[::-1]
    order = np.ascontiguousarray(order, np.int64)
    num = boxes.shape[0]
    keep = np.empty(max(num, 1), np.int64)
    margin = ctypes.c_double(0)
    nk = lib().oracle_nms_rotate(num, _p(boxes), _p(order), ctypes.c_double(iou_threshold),
                                 int(max_output_size), _p(keep), ctypes.byref(margin))
    keep = keep[:nk].copy()
    if return_margin:
        return keep, margin.value
    return keep


def bev_iou(d1, d2):
    """get_iou of test/test_frustum.py:185-189 on rows [x, z, l, w, ry]."""
    d1 = np.ascontiguousarray(d1, np.float64)
    d2 = np.ascontiguousarray(d2, np.float64)
    return lib().oracle_bev_iou(_p(d1), _p(d2))


def nms_on_bev(det, scores, iou_threshold=0.1, return_margin=False):
    """test/test_frustum.py:191-218 for one frame: det (n,5) rows [x, z, l, w, ry] ->
    (group id per detection, kept detection per group)."""
    det = np.ascontiguousarray(det, np.float64).reshape(-1, 5)
    scores = np.ascontiguousarray(scores, np.float64)
    n = det.shape[0]
    group = np.empty(max(n, 1), np.int32)
    keep = np.empty(max(n, 1), np.int32)
    margin = ctypes.c_double(0)
    ng = lib().oracle_bev_group_keep(n, _p(det), _p(scores), ctypes.c_double(iou_threshold), _p(group), _p(keep),
                                     ctypes.byref(margin))
    out = (group[:n].copy(), keep[:ng].copy())
    return out + (margin.value,) if return_margin else out


def mlp_layer(x, w, scale, shift, relu=True):
    x, w = _f32(x), _f32(w)
    scale, shift = _f32(scale), _f32(shift)
    rows = int(np.prod(x.shape[:-1]))
    cin, cout = w.shape
    out = np.empty(x.shape[:-1] + (cout,), np.float32)
    lib().oracle_mlp_layer(ctypes.c_longlong(rows), cin, cout, _p(x), _p(w),
                           _p(scale), _p(shift), int(relu), _p(out))
    return out


def max_pool(x, nsample):
    """x: (..., nsample, c) -> (..., c)"""
    x = _f32(x)
    c = x.shape[-1]
    groups = int(np.prod(x.shape[:-2]))
    out = np.empty(x.shape[:-2] + (c,), np.float32)
    lib().oracle_max_pool(ctypes.c_longlong(groups), nsample, c, _p(x), _p(out))
    return out
