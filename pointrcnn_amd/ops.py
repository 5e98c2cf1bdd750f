"""Reference call surface of the PointNet++ primitive ops on torch-ROCm tensors.

Same names, argument order and result shapes as the reference's TF wrappers:
  models/tf_ops/sampling/tf_sampling.py:30-57      gather_point, farthest_point_sample
  models/tf_ops/grouping/tf_grouping.py:9-42        query_ball_point, group_point
  models/tf_ops/3d_interpolation/tf_interpolate.py:9-29   three_nn, three_interpolate
Tensors are containers only: every op is one call into libpointrcnn_hip.so on the
current torch stream (no eager fallback; a missing library or a CPU tensor raises).
Shape errors raise ValueError, mirroring the OP_REQUIRES checks of the reference's
tf_*.cpp shims.
"""
import os

import torch

from . import _lib

PRECISION_FP32 = 0
PRECISION_BF16 = 1


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (used by bench.py
    to measure each kernel's duration inside the timed region).  Each record carries the
    launch's ALGORITHMIC work (bytes / flops / tests, formulas of SURVEY.md 8d)."""

    def __init__(self):
        self.records = []

    def run(self, name, work, call):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        rc = call()
        e.record()
        self.records.append((name, work, s, e))
        return rc

    def summary(self, by_shape=False):
        """Per kernel family (or, with by_shape, per (family, identical work) = per shape).
        An interval between two events also holds whatever the HOST did between recording the first
        and launching the kernel (a garbage collection, a page fault: tens of ms on an idle stream), so a
        sample above 3x the median of its own shape is replaced by that median (`stalled` counts them)."""
        torch.cuda.synchronize()
        shape_of = lambda name, work: (name, tuple(sorted((k, v) for k, v in work.items() if not callable(v))))
        times = {}
        for name, work, s, e in self.records:
            times.setdefault(shape_of(name, work), []).append(s.elapsed_time(e))
        median = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        out = {}
        for name, work, s, e in self.records:
            shape = shape_of(name, work)
            key = shape if by_shape else name
            d = out.setdefault(key, {"launches": 0, "ms": 0.0, "stalled": 0})
            d["launches"] += 1
            t = s.elapsed_time(e)
            if len(times[shape]) >= 3 and t > 3.0 * median[shape]:
                t = median[shape]
                d["stalled"] += 1
            d["ms"] += t
            for k, v in work.items():
                if callable(v):          # evaluated after the launch finished (data-dependent work)
                    v = v()
                if isinstance(v, str):
                    d[k] = v
                else:
                    d[k] = d.get(k, 0) + v
        return out


_TIMER = None


def set_timer(timer):
    """Install (or with None remove) a KernelTimer; returns the previous one."""
    global _TIMER
    prev, _TIMER = _TIMER, timer
    return prev


def _run(name, work, call):
    if _TIMER is None:
        return call()
    return _TIMER.run(name, work, call)


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(t, dtype, name, ndim=None, last=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise _lib.PointRCNNHipError("%s must live on the GPU (no CPU path exists)" % name)
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s expects a %d-D tensor, got shape %s" % (name, ndim, tuple(t.shape)))
    if last is not None and t.shape[-1] != last:
        raise ValueError("%s expects last dimension %d, got shape %s" % (name, last, tuple(t.shape)))
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


# ---- sampling ---------------------------------------------------------------------------

# PRCNN_FPS_SHARED=1: long chains over 4096 < n <= 16384 points (the first RPN level) run on the
# small-footprint sampler (prcnn_farthest_point_sample_shared: same indices, the cloud in L2 instead
# of one CU's registers), for callers that keep other kernels in flight beside the chain.
FPS_SHARED = os.environ.get("PRCNN_FPS_SHARED", "0") == "1"
FPS_SHARED_MIN_N = int(os.environ.get("PRCNN_FPS_SHARED_MIN_N", "4097"))


def farthest_point_sample(npoint, inp, check=False, shared=None):
    """tf_sampling.py:49-57.  inp (b,n,3) f32 -> (b,npoint) int32.
    shared=True (default: FPS_SHARED, for 4096 < n <= 16384 and npoint >= 256): the sampler that
    leaves its CU to other kernels (n <= 16384), bit-identical indices.
    Clouds of more than 16384 points are sampled by several cooperating workgroups, which can
    fail (bounded wait for a partner that never became resident: see prcnn_farthest_point_sample);
    the result then carries `fps_error`, a 1-element int32 GPU tensor that is 0 on success (an
    asynchronous flag: reading it synchronises), and the unsampled tail of the row is -1.
    check=True reads the flag (one host sync) and raises PointRCNNHipError if it is set."""
    inp = _dev(inp, torch.float32, "FarthestPointSample inp", 3, 3)   # tf_sampling.cpp:105
    b, n, _ = inp.shape
    npoint = int(npoint)
    out = torch.empty((b, max(npoint, 0)), dtype=torch.int32, device=inp.device)
    L = _lib.lib()
    if shared is None:
        shared = FPS_SHARED and FPS_SHARED_MIN_N <= n <= 16384 and npoint >= 256
    if shared:
        tb = L.prcnn_fps_shared_temp_bytes(b, n)
        temp = torch.empty(max(tb, 16), dtype=torch.uint8, device=inp.device)
        _lib.check(_run("fps", {"iters": max(npoint - 1, 0), "dist_evals": b * max(npoint - 1, 0) * n,
                                 "bytes": b * (12 * n + 4 * npoint)},
                        lambda: L.prcnn_farthest_point_sample_shared(b, n, npoint, _p(inp), _p(temp), tb, _p(out),
                                                                     _stream())),
                   "FarthestPointSample (shared)")
        return out
    tb = L.prcnn_fps_temp_bytes(b, n)
    temp = torch.empty(tb // 4, dtype=torch.float32, device=inp.device) if tb else None
    _lib.check(_run("fps", {"iters": max(npoint - 1, 0), "dist_evals": b * max(npoint - 1, 0) * n,
                             "bytes": b * (12 * n + 4 * npoint)},
                    lambda: L.prcnn_farthest_point_sample(b, n, npoint, _p(inp), _p(temp), _p(out), _stream())),
               "FarthestPointSample")
    if temp is not None:
        out.fps_error = temp[:1].view(torch.int32)
        if check and int(out.fps_error.item()) != 0:
            raise _lib.PointRCNNHipError(
                "FarthestPointSample: the cooperating workgroups of a cloud timed out waiting for each other "
                "(not co-resident: is another process filling the GPU?); the tail of the result is -1")
    return out


def _gather_point_fwd(inp, idx):
    """tf_sampling.py:30-38.  inp (b,n,3), idx (b,m) -> (b,m,3)."""
    inp = _dev(inp, torch.float32, "GatherPoint inp", 3, 3)            # tf_sampling.cpp:131
    idx = _dev(idx, torch.int32, "GatherPoint idx", 2)                 # tf_sampling.cpp:135
    if idx.shape[0] != inp.shape[0]:
        raise ValueError("GatherPoint expects idx batch %d, got %d" % (inp.shape[0], idx.shape[0]))
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
    _lib.check(_run("gather_point", {"bytes": b * m * 28},
                    lambda: _lib.lib().prcnn_gather_point(b, n, m, _p(inp), _p(idx), _p(out), _stream())),
               "GatherPoint")
    return out


# ---- grouping ---------------------------------------------------------------------------

def query_ball_point(radius, nsample, xyz1, xyz2):
    """tf_grouping.py:9-21.  xyz1 (b,n,3) points, xyz2 (b,m,3) queries ->
    idx (b,m,nsample) int32, pts_cnt (b,m) int32.  Rows the reference leaves unwritten
    (no point in the ball) are zero-filled by the kernel."""
    xyz1 = _dev(xyz1, torch.float32, "QueryBallPoint xyz1", 3, 3)      # tf_grouping.cpp:79
    xyz2 = _dev(xyz2, torch.float32, "QueryBallPoint xyz2", 3, 3)      # tf_grouping.cpp:84
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("QueryBallPoint expects equal batch sizes")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    nsample = int(nsample)
    idx = torch.empty((b, m, max(nsample, 0)), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    _lib.check(_run("query_ball_point", {"tests": b * m * n, "bytes": b * (12 * n + 12 * m + 4 * m * nsample + 4 * m)},
                    lambda: _lib.lib().prcnn_query_ball_point(b, n, m, float(radius), nsample, _p(xyz1), _p(xyz2),
                                                              _p(idx), _p(cnt), _stream())), "QueryBallPoint")
    return idx, cnt


# Per-cloud n*m from which the spatially culled ball query (nn_sorted.hip: identical idx and
# pts_cnt) replaces the index-order scan; None disables it.  Measured (b16, two radii in one
# pass): 16384 x 4096: 152 us vs 289 us; 4096 x 1024: 48 us vs 30 us (the grid build dominates).
BQ_SORTED_MIN_TESTS = 1 << 24


def _check_order(order, b, n, m, what):
    o1 = _dev(order[0], torch.int32, what + " order1", 2)
    o2 = _dev(order[1], torch.int32, what + " order2", 2)
    if tuple(o1.shape) != (b, n) or tuple(o2.shape) != (b, m):
        raise ValueError("%s: order must be (order of xyz1 (b,n), order of xyz2 (b,m)) from grid_order" % what)
    return o1, o2


def grid_order(xyz1, xyz2):
    """Grid orders of two point sets over their joint bounding box (prcnn_grid_order): the first pass of
    every culled search over this pair, computed once.  Returns (order1 (b,n), order2 (b,m)) int32, to be
    passed as `order=` to the searches over (xyz1, xyz2) -- swapped when the roles are swapped."""
    xyz1 = _dev(xyz1, torch.float32, "grid_order xyz1", 3, 3)
    xyz2 = _dev(xyz2, torch.float32, "grid_order xyz2", 3, 3)
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("grid_order expects equal batch sizes")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    o1 = torch.empty((b, n), dtype=torch.int32, device=xyz1.device)
    o2 = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    _lib.check(_run("grid_order", {"bytes": b * 16 * (n + m)},
                    lambda: _lib.lib().prcnn_grid_order(b, n, m, _p(xyz1), _p(xyz2), _p(o1), _p(o2), _stream())),
               "grid_order")
    return o1, o2


def culled_pair(n, m, nsample_list=None):
    """Whether the searches over point sets of n and m points per cloud take the culled (grid) path:
    the ball query of m centroids in n points and the three_nn of n unknown among m known points."""
    bq = (nsample_list is not None and BQ_SORTED_MIN_TESTS is not None and n * m >= BQ_SORTED_MIN_TESTS
          and all(0 < int(ns) <= 64 for ns in nsample_list))
    return bq, _use_sorted_nn(n, m)


def query_ball_point_msg(radius_list, nsample_list, xyz1, xyz2, order=None):
    """All ball queries of one MSG level in a single pass (same results as one
    query_ball_point call per radius).  Returns [(idx, pts_cnt), ...].  order: grid_order(xyz1, xyz2)
    computed beforehand (used by the culled path only)."""
    import ctypes
    xyz1 = _dev(xyz1, torch.float32, "QueryBallPoint xyz1", 3, 3)
    xyz2 = _dev(xyz2, torch.float32, "QueryBallPoint xyz2", 3, 3)
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("QueryBallPoint expects equal batch sizes")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    k = len(radius_list)
    if k > 4:
        return [query_ball_point(r, ns, xyz1, xyz2) for r, ns in zip(radius_list, nsample_list)]
    res = [(torch.empty((b, m, max(int(ns), 0)), dtype=torch.int32, device=xyz1.device),
            torch.empty((b, m), dtype=torch.int32, device=xyz1.device)) for ns in nsample_list]
    radii = (ctypes.c_float * k)(*[float(r) for r in radius_list])
    nss = (ctypes.c_int * k)(*[int(ns) for ns in nsample_list])
    idxs = (ctypes.c_void_p * k)(*[t[0].data_ptr() for t in res])
    cnts = (ctypes.c_void_p * k)(*[t[1].data_ptr() for t in res])
    L = _lib.lib()
    if (BQ_SORTED_MIN_TESTS is not None and n * m >= BQ_SORTED_MIN_TESTS
            and all(0 < int(ns) <= 64 for ns in nsample_list)):
        nb = L.prcnn_query_ball_point_sorted_workspace_bytes(b, n, m)
        ws = torch.empty((nb,), dtype=torch.uint8, device=xyz1.device)
        o1, o2 = _check_order(order, b, n, m, "QueryBallPoint") if order is not None else (None, None)
        call = lambda: L.prcnn_query_ball_point_msg_ordered(b, n, m, k, radii, nss, _p(xyz1), _p(xyz2), _p(o1), _p(o2),
                                                            idxs, cnts, _p(ws), nb, _stream())
    else:
        call = lambda: L.prcnn_query_ball_point_msg(b, n, m, k, radii, nss, _p(xyz1), _p(xyz2), idxs, cnts, _stream())
    _lib.check(_run("query_ball_point", {"tests": b * m * n,
                                         "bytes": b * (12 * n + 12 * m + sum(4 * m * int(ns) + 4 * m for ns in nsample_list))},
                    call), "QueryBallPoint")
    return res


def select_top_k(k, dist):
    """tf_grouping.py:23-32 (op SelectionSort).  dist (b,m,n) -> idx (b,m,n) int32,
    dist_out (b,m,n): the rows after k selection-sort steps (first k = k smallest)."""
    dist = _dev(dist, torch.float32, "SelectionSort dist", 3)                 # tf_grouping.cpp:118
    b, m, n = dist.shape
    if int(k) <= 0:
        raise ValueError("SelectionSort expects positive k")                  # tf_grouping.cpp:113
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dist.device)
    _lib.check(_run("selection_sort", {"bytes": b * m * n * 12},
                    lambda: _lib.lib().prcnn_selection_sort(b, n, m, int(k), _p(dist), _p(outi), _p(out), _stream())),
               "SelectionSort")
    return outi, out


def knn_point(k, xyz1, xyz2):
    """tf_grouping.py:49-74.  xyz1 (b,n,3) dataset, xyz2 (b,m,3) queries -> val (b,m,k)
    squared distances, idx (b,m,k) int32 (fused: no (b,m,n) matrix)."""
    xyz1 = _dev(xyz1, torch.float32, "knn_point xyz1", 3, 3)
    xyz2 = _dev(xyz2, torch.float32, "knn_point xyz2", 3, 3)
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("knn_point expects equal batch sizes")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    k = int(k)
    if k <= 0:
        raise ValueError("SelectionSort expects positive k")
    val = torch.empty((b, m, k), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, m, k), dtype=torch.int32, device=xyz1.device)
    _lib.check(_run("knn_point", {"tests": b * m * n * 5},
                    lambda: _lib.lib().prcnn_knn_point(b, n, m, k, _p(xyz1), _p(xyz2), _p(val), _p(idx), _stream())),
               "knn_point")
    return val, idx


def _group_point_fwd(points, idx):
    """tf_grouping.py:34-42.  points (b,n,c), idx (b,m,nsample) -> (b,m,nsample,c)."""
    points = _dev(points, torch.float32, "GroupPoint points", 3)       # tf_grouping.cpp:147
    idx = _dev(idx, torch.int32, "GroupPoint idx", 3)                  # tf_grouping.cpp:153
    if idx.shape[0] != points.shape[0]:
        raise ValueError("GroupPoint expects equal batch sizes")
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = torch.empty((b, m, ns, c), dtype=torch.float32, device=points.device)
    _lib.check(_run("group_point", {"bytes": b * m * ns * (4 + 8 * c)},
                    lambda: _lib.lib().prcnn_group_point(b, n, c, m, ns, _p(points), _p(idx), _p(out), _stream())),
               "GroupPoint")
    return out


# ---- 3d_interpolation -------------------------------------------------------------------

# Per-cloud n*m from which the spatially culled search (nn_sorted.hip: bit-identical results)
# is used instead of the brute-force kernels; None disables it.  Measured (b16, fused with the
# interpolation): 16384 x 4096: 217 us vs 355 us brute force; 4096 x 1024: 115 us vs 140 us;
# below that the grid build (one 50 us launch on 2 workgroups per cloud) does not pay.
NN_SORTED_MIN_TESTS = 1 << 22
NN_SORTED_MIN_M = 256


def _use_sorted_nn(n, m):
    return NN_SORTED_MIN_TESTS is not None and n * m >= NN_SORTED_MIN_TESTS and m >= NN_SORTED_MIN_M


def three_nn(xyz1, xyz2, order=None):
    """tf_interpolate.py:9-18.  xyz1 (b,n,3) unknown, xyz2 (b,m,3) known ->
    dist (b,n,3) f32 SQUARED distances, idx (b,n,3) int32.  order: grid_order(xyz1, xyz2) computed
    beforehand (used by the culled path only)."""
    xyz1 = _dev(xyz1, torch.float32, "ThreeNN xyz1", 3, 3)             # tf_interpolate.cpp:163
    xyz2 = _dev(xyz2, torch.float32, "ThreeNN xyz2", 3, 3)             # tf_interpolate.cpp:168
    if xyz1.shape[0] != xyz2.shape[0]:
        raise ValueError("ThreeNN expects equal batch sizes")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    L = _lib.lib()
    if _use_sorted_nn(n, m):
        nb = L.prcnn_three_nn_sorted_workspace_bytes(b, n, m)
        ws = torch.empty((nb,), dtype=torch.uint8, device=xyz1.device)
        o1, o2 = _check_order(order, b, n, m, "ThreeNN") if order is not None else (None, None)
        call = lambda: L.prcnn_three_nn_ordered(b, n, m, _p(xyz1), _p(xyz2), _p(o1), _p(o2), _p(dist), _p(idx), _p(ws), nb,
                                                _stream())
    else:
        call = lambda: L.prcnn_three_nn(b, n, m, _p(xyz1), _p(xyz2), _p(dist), _p(idx), _stream())
    _lib.check(_run("three_nn", {"tests": b * n * m, "bytes": b * (12 * n + 12 * m + 24 * n)}, call), "ThreeNN")
    return dist, idx


def _three_interpolate_fwd(points, idx, weight):
    """tf_interpolate.py:20-29.  points (b,m,c), idx/weight (b,n,3) -> (b,n,c)."""
    points = _dev(points, torch.float32, "ThreeInterpolate points", 3)          # :203
    idx = _dev(idx, torch.int32, "ThreeInterpolate idx", 3, 3)                  # :206
    weight = _dev(weight, torch.float32, "ThreeInterpolate weight", 3, 3)
    if idx.shape != weight.shape or idx.shape[0] != points.shape[0]:
        raise ValueError("ThreeInterpolate expects idx/weight of equal shape (b,n,3)")
    b, m, c = points.shape
    n = idx.shape[1]
    out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
    _lib.check(_run("three_interpolate", {"bytes": b * n * (24 + 16 * c)},
                    lambda: _lib.lib().prcnn_three_interpolate(b, m, c, n, _p(points), _p(idx), _p(weight),
                                                               _p(out), _stream())), "ThreeInterpolate")
    return out


def fp_weights(dist):
    """models/pointnet_util.py:211-214: inverse-distance weights from three_nn's dist."""
    dist = _dev(dist, torch.float32, "fp_weights dist", None, 3)
    w = torch.empty_like(dist)
    _lib.check(_lib.lib().prcnn_fp_weights(dist.numel() // 3, _p(dist), _p(w), _stream()), "fp_weights")
    return w


def three_nn_interpolate(xyz1, xyz2, points2, out=None, return_nn=False):
    """Fused pointnet_util.py:210-215.  Writes (b,n,c) into `out[..., :c]` when a
    wider (b,n,ldo) float32 buffer is given (the FP concat buffer of :218)."""
    xyz1 = _dev(xyz1, torch.float32, "three_nn_interpolate xyz1", 3, 3)
    xyz2 = _dev(xyz2, torch.float32, "three_nn_interpolate xyz2", 3, 3)
    points2 = _dev(points2, torch.float32, "three_nn_interpolate points2", 3)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    c = points2.shape[2]
    if xyz2.shape[0] != b or points2.shape[0] != b or points2.shape[1] != m:
        raise ValueError("three_nn_interpolate expects xyz2 (b,m,3) and points2 (b,m,c)")
    if out is None:
        out = torch.empty((b, n, c), dtype=torch.float32, device=xyz1.device)
    if (not out.is_cuda or out.dtype != torch.float32 or not out.is_contiguous()
            or out.shape[:2] != (b, n) or out.shape[2] < c):
        raise ValueError("three_nn_interpolate: out must be a contiguous (b,n,>=c) float32 GPU tensor")
    dist = idx = None
    if return_nn:
        dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
        idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    L = _lib.lib()
    if _use_sorted_nn(n, m):
        nb = L.prcnn_three_nn_sorted_workspace_bytes(b, n, m)
        ws = torch.empty((nb,), dtype=torch.uint8, device=xyz1.device)
        call = lambda: L.prcnn_three_nn_interpolate_sorted(b, n, m, c, _p(xyz1), _p(xyz2), _p(points2), _p(out),
                                                           out.shape[2], _p(dist), _p(idx), _p(ws), nb, _stream())
    else:
        call = lambda: L.prcnn_three_nn_interpolate(b, n, m, c, _p(xyz1), _p(xyz2), _p(points2), _p(out),
                                                    out.shape[2], _p(dist), _p(idx), _stream())
    _lib.check(_run("three_nn_interpolate", {"tests": b * n * m, "bytes": b * (12 * n + 12 * m) + b * n * 16 * c},
                    call), "three_nn_interpolate")
    return (out, dist, idx) if return_nn else out


# Up to this many queries (b*n) three_nn_interpolate_add runs search and interpolation as
# separate launches.  Measured (b16, fused -> two-step): n256 C512: 61 -> 31 us; n1024 C512:
# 118 -> 48 us; n4096 C256: 119 -> 110 us; n16384 C128: 203 -> 238 us (stays fused).
FP_TWO_STEP_MAX_QUERIES = 65536


def three_nn_interpolate_add(xyz1, xyz2, points2, out, relu=True):
    """out[..., :c] = act(out[..., :c] + interpolate(points2)) in place: epilogue of the split
    FP layer (see prcnn_three_nn_interpolate_add)."""
    xyz1 = _dev(xyz1, torch.float32, "three_nn_interpolate_add xyz1", 3, 3)
    xyz2 = _dev(xyz2, torch.float32, "three_nn_interpolate_add xyz2", 3, 3)
    points2 = _dev(points2, torch.float32, "three_nn_interpolate_add points2", 3)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    c = points2.shape[2]
    if xyz2.shape[0] != b or points2.shape[0] != b or points2.shape[1] != m:
        raise ValueError("three_nn_interpolate_add expects xyz2 (b,m,3) and points2 (b,m,c)")
    if (not isinstance(out, torch.Tensor) or not out.is_cuda or out.dtype != torch.float32
            or not out.is_contiguous() or out.dim() != 3 or out.shape[:2] != (b, n) or out.shape[2] < c):
        raise ValueError("three_nn_interpolate_add: out must be a contiguous (b,n,>=c) float32 GPU tensor")
    L = _lib.lib()
    if FP_TWO_STEP_MAX_QUERIES is not None and b * n <= FP_TWO_STEP_MAX_QUERIES:
        # few queries: the fused kernel's interpolation phase (one workgroup per 256 queries)
        # cannot fill the chip; search, then an element-parallel interpolation
        dist, idx = three_nn(xyz1, xyz2)
        return three_interpolate_add(points2, idx, fp_weights(dist), out, relu)
    ws, nb = None, 0
    if _use_sorted_nn(n, m):
        nb = L.prcnn_three_nn_sorted_workspace_bytes(b, n, m)
        ws = torch.empty((nb,), dtype=torch.uint8, device=xyz1.device)
    _lib.check(_run("three_nn_interpolate", {"tests": b * n * m, "bytes": b * (12 * n + 12 * m) + b * n * 20 * c},
                    lambda: L.prcnn_three_nn_interpolate_add(b, n, m, c, _p(xyz1), _p(xyz2), _p(points2), _p(out),
                                                             out.shape[2], 1 if relu else 0, _p(ws), nb, _stream())),
               "three_nn_interpolate_add")
    return out


def three_interpolate_add(points2, idx, weight, out, relu=True):
    """out[..., :c] = act(out[..., :c] + three_interpolate(points2, idx, weight)) in place, with
    idx / weight (b,n,3) computed beforehand (three_nn + fp_weights)."""
    points2 = _dev(points2, torch.float32, "three_interpolate_add points2", 3)
    idx = _dev(idx, torch.int32, "three_interpolate_add idx", 3, 3)
    weight = _dev(weight, torch.float32, "three_interpolate_add weight", 3, 3)
    b, m, c = points2.shape
    n = idx.shape[1]
    if idx.shape != weight.shape or idx.shape[0] != b:
        raise ValueError("three_interpolate_add expects idx/weight of equal shape (b,n,3)")
    if (not isinstance(out, torch.Tensor) or not out.is_cuda or out.dtype != torch.float32
            or not out.is_contiguous() or out.dim() != 3 or out.shape[:2] != (b, n) or out.shape[2] < c):
        raise ValueError("three_interpolate_add: out must be a contiguous (b,n,>=c) float32 GPU tensor")
    _lib.check(_run("three_interpolate", {"bytes": b * n * (24 + 20 * c)},
                    lambda: _lib.lib().prcnn_three_interpolate_add(b, m, c, n, _p(points2), _p(idx), _p(weight),
                                                                   _p(out), out.shape[2], 1 if relu else 0,
                                                                   _stream())), "three_interpolate_add")
    return out


def three_interpolate_affine(points2, idx, weight, p1, u, t, relu=True):
    """act(three_interpolate(points2, idx, weight) + p1 . u + t) -> (b,n,c): three_nn_interpolate_affine
    with the search done beforehand (same arithmetic).  p1 (b,n,c1<=4) or None, u (c1,c), t (c)."""
    points2 = _dev(points2, torch.float32, "three_interpolate_affine points2", 3)
    idx = _dev(idx, torch.int32, "three_interpolate_affine idx", 3, 3)
    weight = _dev(weight, torch.float32, "three_interpolate_affine weight", 3, 3)
    t = _dev(t, torch.float32, "three_interpolate_affine t", 1)
    b, m, c = points2.shape
    n = idx.shape[1]
    if idx.shape != weight.shape or idx.shape[0] != b or t.shape[0] != c:
        raise ValueError("three_interpolate_affine expects idx/weight (b,n,3) and t (c)")
    c1 = 0
    if p1 is not None:
        p1 = _dev(p1, torch.float32, "three_interpolate_affine p1", 3)
        u = _dev(u, torch.float32, "three_interpolate_affine u", 2)
        c1 = p1.shape[2]
        if p1.shape[:2] != (b, n) or tuple(u.shape) != (c1, c) or c1 > 4:
            raise ValueError("three_interpolate_affine expects p1 (b,n,c1<=4) and u (c1,c)")
    out = torch.empty((b, n, c), dtype=torch.float32, device=points2.device)
    _lib.check(_run("three_interpolate", {"bytes": b * n * (24 + 16 * c + 4 * c1)},
                    lambda: _lib.lib().prcnn_three_interpolate_affine(
        b, m, c, n, _p(points2), _p(idx), _p(weight), _p(p1) if c1 else None, c1, _p(u) if c1 else None, _p(t),
        _p(out), c, 1 if relu else 0, _stream())), "three_interpolate_affine")
    return out


def three_nn_interpolate_affine_supported(b, n, c1):
    """The affine epilogue exists in the fused search+interpolation kernels only (many queries)."""
    return c1 <= 4 and not (FP_TWO_STEP_MAX_QUERIES is not None and b * n <= FP_TWO_STEP_MAX_QUERIES)


def three_nn_interpolate_affine(xyz1, xyz2, points2, p1, u, t, relu=True, order=None):
    """act(interpolate(points2) + p1 . u + t) -> (b,n,c) in one launch: the split first FP layer
    whose points1 half is narrow (p1 (b,n,c1) with c1 <= 4, u (c1,c), t (c); p1 None: t only).
    order: grid_order(xyz1, xyz2) computed beforehand (used by the culled path only)."""
    xyz1 = _dev(xyz1, torch.float32, "three_nn_interpolate_affine xyz1", 3, 3)
    xyz2 = _dev(xyz2, torch.float32, "three_nn_interpolate_affine xyz2", 3, 3)
    points2 = _dev(points2, torch.float32, "three_nn_interpolate_affine points2", 3)
    t = _dev(t, torch.float32, "three_nn_interpolate_affine t", 1)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    c = points2.shape[2]
    if xyz2.shape[0] != b or points2.shape[0] != b or points2.shape[1] != m or t.shape[0] != c:
        raise ValueError("three_nn_interpolate_affine expects xyz2 (b,m,3), points2 (b,m,c), t (c)")
    c1 = 0
    if p1 is not None:
        p1 = _dev(p1, torch.float32, "three_nn_interpolate_affine p1", 3)
        u = _dev(u, torch.float32, "three_nn_interpolate_affine u", 2)
        c1 = p1.shape[2]
        if p1.shape[:2] != (b, n) or tuple(u.shape) != (c1, c):
            raise ValueError("three_nn_interpolate_affine expects p1 (b,n,c1) and u (c1,c)")
    if not three_nn_interpolate_affine_supported(b, n, c1):
        raise ValueError("three_nn_interpolate_affine: c1 <= 4 and more than %s queries required" % FP_TWO_STEP_MAX_QUERIES)
    out = torch.empty((b, n, c), dtype=torch.float32, device=xyz1.device)
    L = _lib.lib()
    ws, nb = None, 0
    o1 = o2 = None
    if _use_sorted_nn(n, m):
        nb = L.prcnn_three_nn_sorted_workspace_bytes(b, n, m)
        ws = torch.empty((nb,), dtype=torch.uint8, device=xyz1.device)
        if order is not None:
            o1, o2 = _check_order(order, b, n, m, "three_nn_interpolate_affine")
    _lib.check(_run("three_nn_interpolate", {"tests": b * n * m,
                                             "bytes": b * (12 * n + 12 * m) + b * n * (16 * c + 4 * c1)},
                    lambda: L.prcnn_three_nn_interpolate_affine_ordered(
        b, n, m, c, _p(xyz1), _p(xyz2), _p(o1), _p(o2), _p(points2), _p(p1), c1, _p(u) if c1 else None, _p(t), _p(out), c,
        1 if relu else 0, _p(ws), nb, _stream())), "three_nn_interpolate_affine")
    return out


# ---- shared MLP -------------------------------------------------------------------------

def mlp_layer(x, w, scale, shift, relu=True, pool=1, precision=PRECISION_FP32, out=None, out_col=0):
    """One shared-MLP layer (tf_util.py:120-185 with BN folded) on the matrix cores.
    x (..., cin) -> (..., cout), or (rows/pool, cout) when pool > 1."""
    x = _dev(x, torch.float32, "mlp_layer x")
    w = _dev(w, torch.float32, "mlp_layer w", 2)
    scale = _dev(scale, torch.float32, "mlp_layer scale", 1)
    shift = _dev(shift, torch.float32, "mlp_layer shift", 1)
    cin, cout = w.shape
    if x.shape[-1] != cin:
        raise ValueError("mlp_layer: x has %d channels, w expects %d" % (x.shape[-1], cin))
    rows = x.numel() // cin
    L = _lib.lib()
    if pool > 1 and (pool % 16 != 0 or 128 % pool != 0):
        # group sizes the fused epilogue does not cover: layer, then the pooling kernel
        y = mlp_layer(x, w, scale, shift, relu, 1, precision)
        res = torch.empty((rows // pool, cout), dtype=torch.float32, device=x.device)
        _lib.check(L.prcnn_max_pool(rows // pool, pool, cout, _p(y), cout, _p(res), cout, _stream()),
                   "max_pool")
        return res
    if out is None:
        shape = (rows // pool, cout) if pool > 1 else tuple(x.shape[:-1]) + (cout,)
        res = torch.empty(shape, dtype=torch.float32, device=x.device)
        optr, ldo = res.data_ptr(), cout
    else:
        res = out
        ldo = out.shape[-1]
        optr = out.data_ptr() + 4 * out_col
    _lib.check(_run("mlp_layer", {"shape": "rows%d K%d N%d pool%d" % (rows, cin, cout, pool), "flops": 2 * rows * cin * cout,
                                   "bytes": 4 * (rows * cin + cin * cout + (rows // pool) * cout)},
                    lambda: L.prcnn_mlp_layer(rows, cin, cout, _p(x), cin, _p(w), _p(scale), _p(shift), int(relu),
                                              int(pool), int(precision), optr, ldo, _stream())), "mlp_layer")
    return res


def sa_group_mlp_layer(xyz, feats, new_xyz, idx, w, scale, shift, xyz_first=False, relu=True, pool=1,
                       precision=PRECISION_FP32):
    """group_point + centre + concat + first shared-MLP layer in one launch
    (pointnet_util.py:177-190 MSG / :43-53 SSG).  Returns (b,m,nsample,cout), or
    (b,m,cout) when pool == nsample."""
    xyz = _dev(xyz, torch.float32, "sa_group_mlp_layer xyz", 3, 3)
    new_xyz = _dev(new_xyz, torch.float32, "sa_group_mlp_layer new_xyz", 3, 3)
    idx = _dev(idx, torch.int32, "sa_group_mlp_layer idx", 3)
    b, n, _ = xyz.shape
    _, m, ns = idx.shape
    c = 0
    if feats is not None:
        feats = _dev(feats, torch.float32, "sa_group_mlp_layer feats", 3)
        c = feats.shape[2]
    w = _dev(w, torch.float32, "sa_group_mlp_layer w", 2)
    scale = _dev(scale, torch.float32, "scale", 1)
    shift = _dev(shift, torch.float32, "shift", 1)
    cout = w.shape[1]
    if w.shape[0] != c + 3:
        raise ValueError("sa_group_mlp_layer: w has %d rows, expected %d" % (w.shape[0], c + 3))
    if pool > 1:
        if pool != ns:
            raise ValueError("sa_group_mlp_layer: pool must equal nsample")
        out = torch.empty((b, m, cout), dtype=torch.float32, device=xyz.device)
    else:
        out = torch.empty((b, m, ns, cout), dtype=torch.float32, device=xyz.device)
    rows = b * m * ns
    _lib.check(_run("sa_group_mlp_layer", {"shape": "rows%d K%d N%d pool%d" % (rows, c + 3, cout, pool),
                                           "flops": 2 * rows * (c + 3) * cout,
                                           "bytes": rows * (4 + 4 * (c + 3)) + 4 * (rows // pool) * cout},
                    lambda: _lib.lib().prcnn_sa_group_mlp_layer(
        b, n, m, ns, c, _p(xyz), _p(feats), _p(new_xyz), _p(idx), int(bool(xyz_first)), cout, _p(w),
        _p(scale), _p(shift), int(relu), int(pool), int(precision), _p(out), cout, _stream())),
        "sa_group_mlp_layer")
    return out


def sa_group_add_layer(xyz, new_xyz, idx, table, col, cout, w_xyz, scale, shift, relu=True, pool=1,
                       precision=PRECISION_FP32):
    """First SA layer with the feature half hoisted out of the grouping: `table` (b,n,ctab) holds
    feats . W_feat per point in columns [col, col+cout); this gathers its rows by idx, adds
    (xyz[idx]-new_xyz) . w_xyz (3,cout) and applies scale/shift/ReLU (and the max-pool).
    Returns (b,m,nsample,cout), or (b,m,cout) when pool == nsample."""
    xyz = _dev(xyz, torch.float32, "sa_group_add_layer xyz", 3, 3)
    new_xyz = _dev(new_xyz, torch.float32, "sa_group_add_layer new_xyz", 3, 3)
    idx = _dev(idx, torch.int32, "sa_group_add_layer idx", 3)
    table = _dev(table, torch.float32, "sa_group_add_layer table", 3)
    w_xyz = _dev(w_xyz, torch.float32, "sa_group_add_layer w_xyz", 2)
    scale = _dev(scale, torch.float32, "scale", 1)
    shift = _dev(shift, torch.float32, "shift", 1)
    b, n, _ = xyz.shape
    _, m, ns = idx.shape
    ctab = table.shape[2]
    if table.shape[0] != b or table.shape[1] != n or col < 0 or col + cout > ctab:
        raise ValueError("sa_group_add_layer: table %s does not hold columns [%d,%d) for %d x %d points"
                         % (tuple(table.shape), col, col + cout, b, n))
    if tuple(w_xyz.shape) != (3, cout):
        raise ValueError("sa_group_add_layer: w_xyz must be (3,%d)" % cout)
    if pool > 1:
        if pool != ns:
            raise ValueError("sa_group_add_layer: pool must equal nsample")
        out = torch.empty((b, m, cout), dtype=torch.float32, device=xyz.device)
    else:
        out = torch.empty((b, m, ns, cout), dtype=torch.float32, device=xyz.device)
    rows = b * m * ns
    # memory-bound gather (3-column product aside): declared by its bytes, not flops
    _lib.check(_run("sa_group_add_layer", {"shape": "rows%d N%d pool%d" % (rows, cout, pool),
                                           "bytes": rows * (4 + 12 + 4 * cout) + 4 * (rows // pool) * cout},
                    lambda: _lib.lib().prcnn_sa_group_add_layer(
        b, n, m, ns, _p(xyz), _p(new_xyz), _p(idx), table.data_ptr() + 4 * col, ctab, cout, _p(w_xyz),
        _p(scale), _p(shift), int(relu), int(pool), int(precision), _p(out), cout, _stream())),
        "sa_group_add_layer")
    return out


def ball_classes(pts_cnt):
    """Sort the centroids of one scale into six classes by pts_cnt (<= 1, 2, 4, 8, 16, 32): returns
    (cls_list (6, b*m) int32, cls_count (8,) int32) for the `classes=` argument of the chain kernels,
    which then skip the padded copies of a ball's first hit (identical results)."""
    pts_cnt = _dev(pts_cnt, torch.int32, "ball_classes pts_cnt", 2)
    groups = pts_cnt.numel()
    lst = torch.empty((6, max(groups, 1)), dtype=torch.int32, device=pts_cnt.device)
    cnt = torch.empty((8,), dtype=torch.int32, device=pts_cnt.device)
    _lib.check(_run("ball_classes", {"bytes": 8 * groups},
                    lambda: _lib.lib().prcnn_ball_count_classes(groups, _p(pts_cnt), _p(lst), _p(cnt), _stream())),
               "ball_count_classes")
    return lst, cnt


class CompactRows:
    """Duplicate-free grouped rows of one scale (prcnn_ball_compact_rows): centroid g owns rows
    off[g] .. off[g+1]-1, its max(cnt,1) distinct ball slots."""

    def __init__(self, idx, pts_cnt, n):
        idx = _dev(idx, torch.int32, "compact rows idx", 3)
        pts_cnt = _dev(pts_cnt, torch.int32, "compact rows pts_cnt", 2)
        b, m, ns = idx.shape
        self.groups, self.capacity = b * m, b * m * ns
        dev = idx.device
        self.rowpi = torch.empty((max(self.capacity, 1),), dtype=torch.int32, device=dev)
        self.rowci = torch.empty((max(self.capacity, 1),), dtype=torch.int32, device=dev)
        self.off = torch.empty((self.groups + 1,), dtype=torch.int32, device=dev)
        self.rows = torch.empty((1,), dtype=torch.int32, device=dev)
        _lib.check(_run("compact_rows", {"bytes": self.groups * (8 + 4 * ns)},
                        lambda: _lib.lib().prcnn_ball_compact_rows(b, n, m, ns, _p(idx), _p(pts_cnt), _p(self.rowpi),
                                                                   _p(self.rowci), _p(self.off), _p(self.rows), _stream())),
                   "ball_compact_rows")

    def count(self):
        return int(self.rows.item())                       # host read-back: instrumentation / tests only


def sa_rows_add_layer(cr, xyz, new_xyz, table, col, cout, w_xyz, scale, shift, relu=True, precision=PRECISION_FP32):
    """sa_group_add_layer over the compacted rows `cr`: returns (capacity, cout), rows past cr.rows unwritten."""
    xyz = _dev(xyz, torch.float32, "sa_rows_add_layer xyz", 3, 3)
    new_xyz = _dev(new_xyz, torch.float32, "sa_rows_add_layer new_xyz", 3, 3)
    table = _dev(table, torch.float32, "sa_rows_add_layer table", 3)
    w_xyz = _dev(w_xyz, torch.float32, "sa_rows_add_layer w_xyz", 2)
    scale = _dev(scale, torch.float32, "scale", 1)
    shift = _dev(shift, torch.float32, "shift", 1)
    ctab = table.shape[2]
    if col < 0 or col + cout > ctab or tuple(w_xyz.shape) != (3, cout):
        raise ValueError("sa_rows_add_layer: table / w_xyz do not hold %d columns at %d" % (cout, col))
    out = torch.empty((cr.capacity, cout), dtype=torch.float32, device=xyz.device)
    _lib.check(_run("sa_group_add_layer", {"shape": "rows<=%d N%d compact" % (cr.capacity, cout),
                                           "bytes": (lambda: cr.count() * (8 + 12 + 8 * cout))},
                    lambda: _lib.lib().prcnn_sa_rows_add_layer(
        cr.capacity, _p(cr.rows), _p(cr.rowpi), _p(cr.rowci), _p(xyz), _p(new_xyz), table.data_ptr() + 4 * col, ctab,
        cout, _p(w_xyz), _p(scale), _p(shift), int(relu), int(precision), _p(out), cout, _stream())),
        "sa_rows_add_layer")
    return out


def mlp_layer_n(cr, x, w, scale, shift, relu=True, precision=PRECISION_FP32):
    """mlp_layer over the first cr.rows rows of x (capacity, cin): the count is read on the device."""
    x = _dev(x, torch.float32, "mlp_layer_n x", 2)
    w = _dev(w, torch.float32, "mlp_layer_n w", 2)
    scale = _dev(scale, torch.float32, "scale", 1)
    shift = _dev(shift, torch.float32, "shift", 1)
    rows, cin = x.shape
    cout = w.shape[1]
    if w.shape[0] != cin or rows != cr.capacity:
        raise ValueError("mlp_layer_n: x (%d,%d) / w %s / capacity %d do not match" % (rows, cin, tuple(w.shape), cr.capacity))
    out = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    _lib.check(_run("mlp_layer", {"shape": "rows<=%d K%d N%d compact" % (rows, cin, cout),
                                  "flops": (lambda: 2 * cr.count() * cin * cout), "flops_nominal": 2 * rows * cin * cout},
                    lambda: _lib.lib().prcnn_mlp_layer_n(rows, _p(cr.rows), cin, cout, _p(x), cin, _p(w), _p(scale),
                                                         _p(shift), int(relu), int(precision), _p(out), cout, _stream())),
               "mlp_layer_n")
    return out


def segment_max(cr, x, out, out_col=0):
    """out[g, out_col:out_col+c] = max over the rows of centroid g of x (capacity, c); out (b,m,>=c) or (groups,>=c)."""
    x = _dev(x, torch.float32, "segment_max x", 2)
    c = x.shape[1]
    if not out.is_cuda or out.dtype != torch.float32 or not out.is_contiguous() or out.shape[-1] < out_col + c:
        raise ValueError("segment_max: out must be a contiguous float32 GPU tensor with >= %d columns" % (out_col + c))
    _lib.check(_run("segment_max", {"bytes": (lambda: 4 * c * (cr.count() + cr.groups))},
                    lambda: _lib.lib().prcnn_segment_max(cr.groups, c, _p(x), c, _p(cr.off), out.data_ptr() + 4 * out_col,
                                                         out.shape[-1], _stream())), "segment_max")
    return out


def sa_chain2_supported(mlp, nsample, precision):
    """Shapes the hoisted-scale chain kernel serves (see prcnn_sa_group_add_mlp2_pool)."""
    return (SA_CHAIN2 and len(mlp) == 3 and mlp[0] <= 64 and mlp[1] <= 96
            and mlp[2] <= 128 and nsample in (16, 32))


SA_CHAIN2 = True     # False: hoisted scales always run layer by layer (tests compare the two)


def sa_group_add_mlp2_pool(xyz, new_xyz, idx, table, col, w_xyz, layers, out=None, out_col=0, classes=None):
    """A hoisted scale in one launch: table-gather first layer (see sa_group_add_layer) + two
    shared-MLP layers + max over nsample.  layers = [(W, scale, shift)] * 3; only the scale /
    shift of layers[0] are used (its weights are `table` and w_xyz (3,c1)).  Returns (b,m,c3), or
    writes into out[..., out_col:out_col+c3]."""
    xyz = _dev(xyz, torch.float32, "sa_group_add_mlp2_pool xyz", 3, 3)
    new_xyz = _dev(new_xyz, torch.float32, "sa_group_add_mlp2_pool new_xyz", 3, 3)
    idx = _dev(idx, torch.int32, "sa_group_add_mlp2_pool idx", 3)
    table = _dev(table, torch.float32, "sa_group_add_mlp2_pool table", 3)
    w_xyz = _dev(w_xyz, torch.float32, "sa_group_add_mlp2_pool w_xyz", 2)
    b, n, _ = xyz.shape
    _, m, ns = idx.shape
    ctab = table.shape[2]
    (_, s1, t1), (W2, s2, t2), (W3, s3, t3) = [tuple(_dev(t, torch.float32, "layer") for t in L) for L in layers]
    c1, c2, c3 = w_xyz.shape[1], W2.shape[1], W3.shape[1]
    if table.shape[0] != b or table.shape[1] != n or col < 0 or col + c1 > ctab:
        raise ValueError("sa_group_add_mlp2_pool: table %s does not hold columns [%d,%d) for %d x %d points"
                         % (tuple(table.shape), col, col + c1, b, n))
    if w_xyz.shape[0] != 3 or W2.shape[0] != c1 or W3.shape[0] != c2 or s1.shape[0] != c1:
        raise ValueError("sa_group_add_mlp2_pool: layer shapes do not chain")
    if out is None:
        res = torch.empty((b, m, c3), dtype=torch.float32, device=xyz.device)
        optr, ldo = res.data_ptr(), c3
    else:
        res, ldo, optr = out, out.shape[-1], out.data_ptr() + 4 * out_col
    rows = b * m * ns
    per_row = 2 * (3 * c1 + c1 * c2 + c2 * c3)
    # flops = the work the launch executes (padded copies skipped); flops_nominal = SURVEY.md 8d's formula
    _lib.check(_run("sa_chain2", {"shape": "rows%d N%d-%d-%d pool%d" % (rows, c1, c2, c3, ns),
                                  "flops": (lambda: per_row * _cls_rows(classes, rows)), "flops_nominal": per_row * rows,
                                  "bytes": rows * (4 + 12 + 4 * c1) + 4 * (rows // ns) * c3},
                    (lambda: _lib.lib().prcnn_sa_group_add_mlp2_pool(
        b, n, m, ns, _p(xyz), _p(new_xyz), _p(idx), table.data_ptr() + 4 * col, ctab, c1, c2, c3, _p(w_xyz),
        _p(s1), _p(t1), _p(W2), _p(s2), _p(t2), _p(W3), _p(s3), _p(t3), optr, ldo, _stream()))
                    if classes is None else
                    (lambda: _lib.lib().prcnn_sa_group_add_mlp2_pool_classes(
        b, n, m, ns, _p(xyz), _p(new_xyz), _p(idx), table.data_ptr() + 4 * col, ctab, c1, c2, c3, _p(w_xyz),
        _p(s1), _p(t1), _p(W2), _p(s2), _p(t2), _p(W3), _p(s3), _p(t3), optr, ldo, _p(_cls_check(classes, b * m)[0]),
        _p(classes[1]), _stream()))),
        "sa_group_add_mlp2_pool")
    return res


def sa_chain_supported(c, mlp, nsample, precision):
    """Shapes the fully fused set-abstraction kernel serves (see prcnn_sa_group_mlp3_pool)."""
    return (len(mlp) == 3 and c + 3 <= 8 and mlp[0] <= 32
            and mlp[1] <= 32 and mlp[2] <= 128 and nsample in (16, 32))


def _cls_rows(classes, nominal_rows):
    """Rows a chain kernel evaluates with class lists: whole 32-row tiles of count_k groups of
    1 << k slots (read back from the device: instrumentation only)."""
    if classes is None:
        return nominal_rows
    c = classes[1][:6].tolist()
    return sum(32 * ((n * (1 << k) + 31) // 32) for k, n in enumerate(c))


def _cls_check(classes, groups):
    lst, cnt = classes
    if (not lst.is_cuda or lst.dtype != torch.int32 or not lst.is_contiguous() or tuple(lst.shape) != (6, max(groups, 1))
            or not cnt.is_cuda or cnt.dtype != torch.int32 or cnt.numel() < 6):
        raise ValueError("classes must be the (cls_list (6,%d) int32, cls_count int32) pair of ball_classes" % groups)
    return lst, cnt


def sa_group_mlp3_pool(xyz, feats, new_xyz, idx, layers, xyz_first=False, out=None, out_col=0, classes=None):
    """Grouping + three shared-MLP layers + max over nsample in one launch (narrow layers).
    layers = [(W, scale, shift)] * 3.  Returns (b, m, c3), or writes into out[..., out_col:]."""
    xyz = _dev(xyz, torch.float32, "sa_group_mlp3_pool xyz", 3, 3)
    new_xyz = _dev(new_xyz, torch.float32, "sa_group_mlp3_pool new_xyz", 3, 3)
    idx = _dev(idx, torch.int32, "sa_group_mlp3_pool idx", 3)
    b, n, _ = xyz.shape
    _, m, ns = idx.shape
    c = 0
    if feats is not None:
        feats = _dev(feats, torch.float32, "sa_group_mlp3_pool feats", 3)
        c = feats.shape[2]
    (W1, s1, t1), (W2, s2, t2), (W3, s3, t3) = [tuple(_dev(t, torch.float32, "layer") for t in L) for L in layers]
    c1, c2, c3 = W1.shape[1], W2.shape[1], W3.shape[1]
    if W1.shape[0] != c + 3 or W2.shape[0] != c1 or W3.shape[0] != c2:
        raise ValueError("sa_group_mlp3_pool: layer shapes do not chain")
    if out is None:
        res = torch.empty((b, m, c3), dtype=torch.float32, device=xyz.device)
        optr, ldo = res.data_ptr(), c3
    else:
        res, ldo, optr = out, out.shape[-1], out.data_ptr() + 4 * out_col
    rows = b * m * ns
    per_row = 2 * ((c + 3) * c1 + c1 * c2 + c2 * c3)
    _lib.check(_run("sa_chain", {"shape": "rows%d K%d N%d-%d-%d pool%d" % (rows, c + 3, c1, c2, c3, ns),
                                 "flops": (lambda: per_row * _cls_rows(classes, rows)), "flops_nominal": per_row * rows, "bytes": rows * (4 + 4 * (c + 3)) + 4 * (rows // ns) * c3},
                    (lambda: _lib.lib().prcnn_sa_group_mlp3_pool(
        b, n, m, ns, c, _p(xyz), _p(feats), _p(new_xyz), _p(idx), int(bool(xyz_first)), c1, c2, c3,
        _p(W1), _p(s1), _p(t1), _p(W2), _p(s2), _p(t2), _p(W3), _p(s3), _p(t3), optr, ldo, _stream()))
                    if classes is None else
                    (lambda: _lib.lib().prcnn_sa_group_mlp3_pool_classes(
        b, n, m, ns, c, _p(xyz), _p(feats), _p(new_xyz), _p(idx), int(bool(xyz_first)), c1, c2, c3,
        _p(W1), _p(s1), _p(t1), _p(W2), _p(s2), _p(t2), _p(W3), _p(s3), _p(t3), optr, ldo,
        _p(_cls_check(classes, b * m)[0]), _p(classes[1]), _stream()))),
        "sa_group_mlp3_pool")
    return res


# ---- backward ops ----------------------------------------------------
# Same argument lists as the reference's gradient ops (tf_sampling.py:44-48,
# tf_grouping.py:43-47, tf_interpolate.py:30-35): the forward input comes first and only
# its shape is used.  gather_point / group_point / three_interpolate route through the
# autograd Functions below when their feature input requires grad, which is what
# tf.RegisterGradient does in the reference.

def _workspace(nbytes, device):
    return torch.empty((max(int(nbytes), 4),), dtype=torch.uint8, device=device)


def gather_point_grad(inp, idx, out_g):
    """GatherPointGrad, tf_sampling.cpp:151-177.  inp (b,n,3) [shape only], idx (b,m),
    out_g (b,m,3) -> inp_g (b,n,3)."""
    idx = _dev(idx, torch.int32, "GatherPointGrad idx", 2)
    out_g = _dev(out_g, torch.float32, "GatherPointGrad out_g", 3, 3)
    if inp.dim() != 3 or inp.shape[2] != 3:
        raise ValueError("GatherPointGradGpuOp expects (batch_size,num_points,3) inp")       # tf_sampling.cpp:156
    b, n = int(inp.shape[0]), int(inp.shape[1])
    m = idx.shape[1]
    if idx.shape[0] != b or tuple(out_g.shape) != (b, m, 3):
        raise ValueError("GatherPointGradGpuOp expects (batch_size,num_result) idx and (batch_size,num_result,3) out_g")
    L = _lib.lib()
    nb = L.prcnn_gather_point_grad_workspace_bytes(b, n, m)
    ws = _workspace(nb, out_g.device)
    inp_g = torch.empty((b, n, 3), dtype=torch.float32, device=out_g.device)
    _lib.check(_run("gather_point_grad", {"bytes": b * m * 28 + b * n * 12},
                    lambda: L.prcnn_gather_point_grad(b, n, m, _p(out_g), _p(idx), _p(inp_g), _p(ws), nb, _stream())),
               "GatherPointGrad")
    return inp_g


def group_point_grad(points, idx, grad_out):
    """GroupPointGrad, tf_grouping.cpp:174-207.  points (b,n,c) [shape only], idx (b,m,ns),
    grad_out (b,m,ns,c) -> grad_points (b,n,c)."""
    idx = _dev(idx, torch.int32, "GroupPointGrad idx", 3)
    grad_out = _dev(grad_out, torch.float32, "GroupPointGrad grad_out", 4)
    if points.dim() != 3:
        raise ValueError("GroupPointGrad expects (batch_size, num_points, channel) points shape")   # :180
    b, n, c = (int(v) for v in points.shape)
    _, m, ns = idx.shape
    if idx.shape[0] != b or tuple(grad_out.shape) != (b, m, ns, c):
        raise ValueError("GroupPointGrad expects (b,m,nsample) idx and (b,m,nsample,c) grad_out")
    L = _lib.lib()
    nb = L.prcnn_group_point_grad_workspace_bytes(b, n, m, ns)
    ws = _workspace(nb, grad_out.device)
    g = torch.empty((b, n, c), dtype=torch.float32, device=grad_out.device)
    _lib.check(_run("group_point_grad", {"bytes": b * m * ns * (4 + 4 * c) + b * n * c * 4},
                    lambda: L.prcnn_group_point_grad(b, n, c, m, ns, _p(grad_out), _p(idx), _p(g), _p(ws), nb,
                                                     _stream())),
               "GroupPointGrad")
    return g


def three_interpolate_grad(points, idx, weight, grad_out):
    """ThreeInterpolateGrad, tf_interpolate.cpp:225-261.  points (b,m,c) [shape only],
    idx (b,n,3), weight (b,n,3), grad_out (b,n,c) -> grad_points (b,m,c)."""
    idx = _dev(idx, torch.int32, "ThreeInterpolateGrad idx", 3, 3)
    weight = _dev(weight, torch.float32, "ThreeInterpolateGrad weight", 3, 3)
    grad_out = _dev(grad_out, torch.float32, "ThreeInterpolateGrad grad_out", 3)
    if points.dim() != 3:
        raise ValueError("ThreeInterpolateGrad expects (b,m,c) points shape")                    # :231
    b, m, c = (int(v) for v in points.shape)
    n = idx.shape[1]
    if idx.shape[0] != b or tuple(weight.shape) != (b, n, 3) or tuple(grad_out.shape) != (b, n, c):
        raise ValueError("ThreeInterpolateGrad expects (b,n,3) idx/weight and (b,n,c) grad_out")
    L = _lib.lib()
    nb = L.prcnn_three_interpolate_grad_workspace_bytes(b, n, m)
    ws = _workspace(nb, grad_out.device)
    g = torch.empty((b, m, c), dtype=torch.float32, device=grad_out.device)
    _lib.check(_run("three_interpolate_grad", {"bytes": b * n * (24 + 12 * c) + b * m * c * 4},
                    lambda: L.prcnn_three_interpolate_grad(b, n, c, m, _p(grad_out), _p(idx), _p(weight), _p(g),
                                                           _p(ws), nb, _stream())),
               "ThreeInterpolateGrad")
    return g


class _GatherPointFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        ctx.save_for_backward(idx)
        ctx.inp_shape = inp.shape
        return _gather_point_fwd(inp, idx)

    @staticmethod
    def backward(ctx, out_g):
        (idx,) = ctx.saved_tensors
        return gather_point_grad(torch.empty(ctx.inp_shape, device="meta"), idx, out_g), None


class _GroupPointFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        ctx.save_for_backward(idx)
        ctx.points_shape = points.shape
        return _group_point_fwd(points, idx)

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        return group_point_grad(torch.empty(ctx.points_shape, device="meta"), idx, grad_out), None


class _ThreeInterpolateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        ctx.save_for_backward(idx, weight)
        ctx.points_shape = points.shape
        return _three_interpolate_fwd(points, idx, weight)

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        return (three_interpolate_grad(torch.empty(ctx.points_shape, device="meta"), idx, weight, grad_out),
                None, None)


def _wants_grad(t):
    return isinstance(t, torch.Tensor) and t.requires_grad and torch.is_grad_enabled()


def gather_point(inp, idx):
    """Differentiable front of _gather_point_fwd (same contract; see there)."""
    if _wants_grad(inp):
        return _GatherPointFn.apply(inp, idx)
    return _gather_point_fwd(inp, idx)


def group_point(points, idx):
    """Differentiable front of _group_point_fwd (same contract; see there)."""
    if _wants_grad(points):
        return _GroupPointFn.apply(points, idx)
    return _group_point_fwd(points, idx)


def three_interpolate(points, idx, weight):
    """Differentiable front of _three_interpolate_fwd (same contract; see there)."""
    if _wants_grad(points):
        return _ThreeInterpolateFn.apply(points, idx, weight)
    return _three_interpolate_fwd(points, idx, weight)
