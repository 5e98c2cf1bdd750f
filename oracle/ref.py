"""ctypes access to the reference's OWN standalone CPU sources, compiled by
oracle/Makefile into oracle/_ref/ (TEST INFRASTRUCTURE ONLY).

The libraries are built from the files where they lie under /root/reference:
  models/tf_ops/grouping/query_ball_point.cpp      -> libref_grouping.so
  models/tf_ops/3d_interpolation/interpolate.cpp   -> libref_interp.so
  models/tf_ops/grouping/selection_sort.cpp        -> libref_ssort.so
They are C++ with no extern "C", so the Itanium-mangled names are used.
oracle/_ref is git-ignored (binary), travels to the GPU box with the tree, and is
only used to validate the restatement in oracle/pointrcnn_oracle.c.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")


def available():
    return all(os.path.exists(os.path.join(_REF, f)) for f in
               ("libref_grouping.so", "libref_interp.so"))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


_g = _i = None


def _grouping():
    global _g
    if _g is None:
        _g = ctypes.CDLL(os.path.join(_REF, "libref_grouping.so"))
    return _g


def _interp():
    global _i
    if _i is None:
        _i = ctypes.CDLL(os.path.join(_REF, "libref_interp.so"))
    return _i


def query_ball_point_cpu(radius, nsample, xyz1, xyz2):
    """query_ball_point.cpp:19-47 (no pts_cnt; idx pre-zeroed like its main(), :94)."""
    xyz1 = np.ascontiguousarray(xyz1, np.float32)
    xyz2 = np.ascontiguousarray(xyz2, np.float32)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    _grouping()._Z20query_ball_point_cpuiiifiPKfS0_Pi(
        b, n, m, ctypes.c_float(radius), nsample, _p(xyz1), _p(xyz2), _p(idx))
    return idx


def group_point_cpu(points, idx):
    """query_ball_point.cpp:52-66"""
    points = np.ascontiguousarray(points, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.empty((b, m, ns, c), np.float32)
    _grouping()._Z15group_point_cpuiiiiiPKfPKiPf(b, n, c, m, ns, _p(points), _p(idx), _p(out))
    return out


def interpolate_cpu(points, idx, weight):
    """interpolate.cpp:84-104 (loop-identical to tf_interpolate.cpp:107-127)."""
    points = np.ascontiguousarray(points, np.float32)
    idx = np.ascontiguousarray(idx, np.int32)
    weight = np.ascontiguousarray(weight, np.float32)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    _interp()._Z15interpolate_cpuiiiiPKfPKiS0_Pf(b, m, c, n, _p(points), _p(idx), _p(weight), _p(out))
    return out


_s = None


def _ssort():
    global _s
    if _s is None:
        _s = ctypes.CDLL(os.path.join(_REF, "libref_ssort.so"))
    return _s


def selection_sort_cpu(k, dist):
    """selection_sort.cpp:20-63.  The reference function printf()s every element and every
    pick; file descriptor 1 is pointed at /dev/null for the duration of the call.
    Returns (idx (b,m,n) int32, val (b,m,n) f32)."""
    dist = np.ascontiguousarray(dist, np.float32)
    b, m, n = dist.shape
    idx = np.empty((b, m, n), np.int32)
    val = np.empty((b, m, n), np.float32)
    import sys
    sys.stdout.flush()
    saved = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    try:
        os.dup2(devnull, 1)
        _ssort()._Z18selection_sort_cpuiiiiPKfPiPf(b, n, m, int(k), _p(dist), _p(idx), _p(val))
        ctypes.CDLL(None).fflush(None)
    finally:
        os.dup2(saved, 1)
        os.close(saved)
        os.close(devnull)
    return idx, val


def group_point_grad_cpu(n, idx, grad_out):
    """query_ball_point.cpp:70-84 (accumulates into a caller-cleared buffer)."""
    idx = np.ascontiguousarray(idx, np.int32)
    grad_out = np.ascontiguousarray(grad_out, np.float32)
    b, m, ns, c = grad_out.shape
    g = np.zeros((b, n, c), np.float32)
    _grouping()._Z20group_point_grad_cpuiiiiiPKfPKiPf(b, n, c, m, ns, _p(grad_out), _p(idx), _p(g))
    return g


def interpolate_grad_cpu(m, idx, weight, grad_out):
    """interpolate.cpp:108-130 (loop-identical to tf_interpolate.cpp:131-153)."""
    idx = np.ascontiguousarray(idx, np.int32)
    weight = np.ascontiguousarray(weight, np.float32)
    grad_out = np.ascontiguousarray(grad_out, np.float32)
    b, n, c = grad_out.shape
    g = np.zeros((b, m, c), np.float32)
    _interp()._Z20interpolate_grad_cpuiiiiPKfPKiS0_Pf(b, n, c, m, _p(grad_out), _p(idx), _p(weight), _p(g))
    return g


def threenn_origin_cpu(n, xyz2):
    """interpolate.cpp:21-64: the standalone twin whose distance is
    d = x2*x2+y2*y2+z2*z2 (:34) -- i.e. tf_interpolate.cpp:60-103 evaluated with
    every query point at the origin.  Returns (dist, idx) of shape (b,n,3)."""
    xyz2 = np.ascontiguousarray(xyz2, np.float32)
    b, m, _ = xyz2.shape
    xyz1 = np.zeros((b, n, 3), np.float32)
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    _interp()._Z11threenn_cpuiiiPKfS0_PfPi(b, n, m, _p(xyz1), _p(xyz2), _p(dist), _p(idx))
    return dist, idx
