#!/usr/bin/env python3
"""Benchmark of the PointRCNN point-cloud hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU over RCCL.)

Workload (BASELINE.json configs[2], the config the metric "SA/FP-stack scenes/sec
(16k pts)" is quoted on): the PointRCNN RPN SA+FP stack on a batch of 16 synthetic
KITTI-shaped scenes x 16384 points per GPU, fused group->shared-MLP->max-pool on the
matrix cores, followed by the rotated BEV NMS of each scene's 2048 proposals (top 100,
IoU 0.8: the reference's val/test setting) and ONE all-gather (RCCL for N>1) of all
per-scene detection records produced in the timed region, issued at its closing fence.
One "step" = one such batch per rank (weak scaling).  Inputs are resident in HBM before
the timed region; nothing is allocated inside it.

Schedule (RoundSchedule, the default): three step streams, each replaying one linear HIP
graph per unit of up to 4 batches (the feature phases), beside ONE sampling stream whose
single farthest-point-sampling chain per round samples the clouds of the whole NEXT
round (software pipelining: every batch still gets a complete FPS pass and a complete
feature phase); `round_plan` deals any K into rounds, so exactly K steps are timed and
nothing is issued eagerly.  After the timed region a replay's features and records are
compared bit for bit with an eager, un-pipelined step.  PRCNN_BENCH_ROUNDS=0 or
--fps-group N: the earlier schedule (PipelinedSchedule / Unit: one forked graph per step
stream, features + that unit's own chain; K mod group remainder steps eager).

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant roofline-bound launch;
durations from HIP events in an eager single-stream pass, HBM traffic from the committed
rocprofv3 PMC passes), `kernels` (every op family), `fps` (the latency-bound sampler: us per
iteration + SQ counters), `nms` (ms @ 9000 boxes), `value_no_dedup` (the same timed region
with every padded ball slot evaluated), `single_scene_latency_ms`, `frustum`, `fps_dense`,
`cpu_baseline` (the CPU oracle = reference CPU path: 1 core, and farmed over 16 processes).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from pointrcnn_amd import dist as pdist
from pointrcnn_amd import ops, pointnet_util as PU, roofline, stacks, synthetic as S
from pointrcnn_amd.nms_rotate import nms_rotate_batched, nms_rotate_padded

BATCH = 16
NPTS = 16384
NPROP = 2048          # NUM_FG_POINT proposals per frame (models/model_util.py:16)


def make_inputs(rank, device, batch):
    pcs = torch.as_tensor(S.kitti_batch(rank * batch, batch, NPTS)).to(device)
    boxes, scores, boxes3d = [], [], []
    for i in range(batch):
        b, s = S.nms_boxes(seed=7000 + rank * batch + i, N=NPROP, clusters=40)
        boxes.append(b); scores.append(s)
        b3 = np.stack([b[:, 0], np.full(NPROP, 1.0), b[:, 1], b[:, 2], np.full(NPROP, 1.5), b[:, 3],
                       np.deg2rad(b[:, 4])], 1)
        boxes3d.append(b3)
    prop = {
        "bev": torch.as_tensor(np.stack(boxes)).to(device),            # (B,N,5) f64
        "scores": torch.as_tensor(np.stack(scores)).to(device),        # (B,N) f64
        "boxes3d": torch.as_tensor(np.stack(boxes3d)).to(device),      # (B,N,7) f64
        "scene_ids": torch.arange(rank * batch, (rank + 1) * batch, dtype=torch.float32, device=device),
    }
    return pcs, prop


def detections(prop):
    """Per-scene rotated NMS of the batch's proposals (2048 -> 100, IoU 0.8) + detection records."""
    keep, nk = nms_rotate_batched(prop["bev"], prop["scores"], 0.8, pdist.MAX_DET)
    return pdist.pack_detections(prop["scene_ids"], prop["boxes3d"], prop["scores"], keep, nk)


def step(pcs, prop, store, precision, geo=None, centroids=None):
    """SA+FP stack on the batch, per-scene rotated NMS of its proposals, detection records."""
    feats, _ = stacks.rpn_sa_fp_stack(pcs, params=store, precision=precision, geometry_stream=geo,
                                      centroids=centroids)
    return feats, detections(prop)


_SPIN = {}


def _spin(blocks, ms, mode, device):
    """Probe only: tools/micro/libspin.so's resident kernel (one 256-thread workgroup per cloud for `ms`)."""
    import ctypes
    if not _SPIN:
        lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "micro", "libspin.so"))
        lib.spin_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p]
        _SPIN["lib"] = lib
        _SPIN["ring"] = torch.randint(0, 1 << 20, (1 << 20,), dtype=torch.int32, device=device)
        _SPIN["sink"] = torch.zeros(2 * (128 + 1024 * 512), dtype=torch.int32, device=device)
    rc = _SPIN["lib"].spin_launch(blocks, ms, mode, _SPIN["ring"].data_ptr(), _SPIN["ring"].numel(),
                                  _SPIN["sink"].data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


def ahead_work(pcs_next, cent_next, group):
    """What runs AHEAD of (and beside) the feature phases: ONE strictly serial FPS chain over the
    clouds of the next `group` batches into `cent_next`."""
    xyz_next = pcs_next[:, :, 0:3].contiguous()
    if group > 1:
        xyz_next = xyz_next.repeat(group, 1, 1)              # synthetic input: the same batch again
    npoints = [lv[0] for lv in stacks.RPN_SA]
    # The chain runs beside the feature phases of two streams: level 1 on the sampler that leaves its
    # CU to them (prcnn_farthest_point_sample_shared; PRCNN_BENCH_FPS_SHARED=0: the register-resident one).
    shared = os.environ.get("PRCNN_BENCH_FPS_SHARED", "1") != "0"
    probe = os.environ.get("PRCNN_BENCH_PROBE_NO_FPS")        # probes only (not valid bench lines)
    if probe in ("70", "71", "72", "73", "74", "75"):               # level 1 replaced by a resident do-nothing / load / VALU / load+store kernel
        if probe == "74":                                     # ... preceded by the shared sampler's sort launch alone (npoint = 1)
            ops.farthest_point_sample(1, xyz_next, shared=True)
        _spin(xyz_next.shape[0], 4.0, {"74": 0, "75": 4}.get(probe, int(probe) - 70), xyz_next.device)
        probe = "2"
    if probe == "2":                                          # levels 2.. only (level 1 keeps its centroids)
        xs = [cent_next[0]] + stacks.sample_centroid_chain(cent_next[0], npoints[1:], shared=shared)[0]
    elif probe == "4":                                        # the fork / join and the copies alone: no sampling
        xs = [c + 0.0 for c in cent_next]
    elif probe == "5":                                        # an (almost) empty side branch
        cent_next[-1].add_(0.0)
        return
    elif probe == "3":                                        # level 1 only
        xs = stacks.sample_centroid_chain(xyz_next, npoints[:1], shared=shared)[0]
    else:
        xs, _ = stacks.sample_centroid_chain(xyz_next, npoints, shared=shared)
    for dst, x in zip(cent_next, xs):
        if dst is not x:
            dst.copy_(x)


def feature_work(pcs_cur, prop, store, precision, cent_cur, group):
    """The feature phases (grouping, MLPs, interpolation, NMS + records) of the current `group`
    batches with the centroids prepared one replay earlier."""
    b = pcs_cur.shape[0]
    return [step(pcs_cur, prop, store, precision, centroids=[c[g * b:(g + 1) * b] for c in cent_cur])
            for g in range(group)]


def pipelined_step(pcs_cur, pcs_next, prop, store, precision, cent_cur, cent_next, side, group=1):
    """`group` steps of the software-pipelined schedule as ONE unit of work forked over two
    streams: feature_work of the current `group` batches on the current stream (centroids from
    `cent_cur`), ahead_work for the next `group` batches on `side` (into `cent_next`); after the
    join `cent_next` is copied into `cent_cur` (4 MB per 4 batches), so ONE unit serves every replay
    instead of a ping-pong of two.  Every batch still gets one complete FPS pass and one complete
    feature phase; only their pairing is shifted, so the latency-bound sampling (one chain costs the
    same whatever the number of clouds, 16 CUs per batch) never gates full-chip work, and one
    chain's latency is spread over `group` batches.  Returns one (feats, rec) per batch."""
    cur = torch.cuda.current_stream()
    probe = os.environ.get("PRCNN_BENCH_PROBE_NO_FPS")        # probes only (not valid bench lines)
    if probe == "1":                                          # feature phases alone: no fork, no copies
        return feature_work(pcs_cur, prop, store, precision, cent_cur, group)
    if probe == "6":                                          # no fork; the copies on the main stream
        outs = feature_work(pcs_cur, prop, store, precision, cent_cur, group)
        for dst, src in zip(cent_cur, cent_next):
            dst.copy_(src)
        return outs
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        ahead_work(pcs_next, cent_next, group)
    outs = feature_work(pcs_cur, prop, store, precision, cent_cur, group)
    cur.wait_stream(side)
    if probe != "5":
        for dst, src in zip(cent_cur, cent_next):
            dst.copy_(src)
    return outs


class Unit:
    """One replayable unit of the pipelined schedule on one stream: `group` batches, as HIP graphs
    (graph=True) or issued eagerly.  The unit OWNS every tensor its graphs read or write (the
    centroid sets, the outputs) for as long as the graphs live.
    PRCNN_BENCH_SPLIT=1: the feature phases and the sampling chain as TWO linear graphs, replayed on
    the step stream and its companion stream and joined by stream events, instead of ONE graph that
    holds the fork and the join.  Measured: both forms reach the same rate when every
    one of the four lanes (2 x features, 2 x chain) has a hardware queue to itself -- the single graph
    with HIP's default 4 queues, the two graphs with 8 -- and both lose a third (8.2 - 8.5 k scenes/s)
    with the other queue count, where a chain shares a queue with the other stream's features."""

    def __init__(self, st, side, pcs, prop, store, precision, group, graph=True):
        self.st, self.side, self.group = st, side, group
        self.args = (pcs, pcs, prop, store, precision)
        self.split = os.environ.get("PRCNN_BENCH_SPLIT", "0") == "1" and os.environ.get("PRCNN_BENCH_PROBE_NO_FPS") not in ("1", "6")
        with torch.cuda.stream(st):
            xyz_rep = pcs[:, :, 0:3].contiguous().repeat(group, 1, 1)
            xs, _ = stacks.sample_centroid_chain(xyz_rep, [lv[0] for lv in stacks.RPN_SA])
            self.cent_cur = [x.clone() for x in xs]
            self.cent_next = [x.clone() for x in xs]
            self.outs = pipelined_step(*self.args, self.cent_cur, self.cent_next, side, group)   # warm-up
        torch.cuda.synchronize()
        self.graph = self.g_chain = self.g_feat = None
        if graph and self.split:
            self.g_chain = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_chain, stream=side):
                ahead_work(pcs, self.cent_next, group)
            self.g_feat = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_feat, stream=st):
                self.outs = feature_work(pcs, prop, store, precision, self.cent_cur, group)
        elif graph:
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=st):
                self.outs = pipelined_step(*self.args, self.cent_cur, self.cent_next, side, group)

    def run(self):
        """Issue the unit on its stream (the caller has made it current)."""
        if self.g_feat is not None:
            cur = torch.cuda.current_stream()
            self.side.wait_stream(cur)              # the previous unit's copy out of cent_next has been issued on cur
            with torch.cuda.stream(self.side):
                self.g_chain.replay()
            self.g_feat.replay()
            cur.wait_stream(self.side)
            if os.environ.get("PRCNN_BENCH_PROBE_NO_FPS") != "5":
                for dst, src in zip(self.cent_cur, self.cent_next):
                    dst.copy_(src)
        elif self.graph is not None:
            self.graph.replay()
        else:
            self.outs = pipelined_step(*self.args, self.cent_cur, self.cent_next, self.side, self.group)
        return self.outs


def ball_fill(pcs):
    """Mean pts_cnt / nsample per SA scale on the bench scenes: the share of grouped rows that are
    distinct (the rest are the padding copies the duplicate-free kernels skip)."""
    xyz = pcs[:, :, 0:3].contiguous()
    out = []
    for npoint, radii, nsamples, _ in stacks.RPN_SA:
        new_xyz = ops.gather_point(xyz, ops.farthest_point_sample(npoint, xyz))
        for (idx, cnt), ns in zip(ops.query_ball_point_msg(radii, nsamples, xyz, new_xyz), nsamples):
            out.append(round(float(cnt.float().mean().item()) / ns, 3))
        xyz = new_xyz
    return out


def time_nms(device, n=9000, iters=20):
    boxes, scores = S.nms_boxes(seed=9000, N=n)
    b = torch.as_tensor(boxes).to(device); s = torch.as_tensor(scores).to(device)
    for _ in range(3):
        nms_rotate_padded(b, s, 0.8, 100)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        k, nk = nms_rotate_padded(b, s, 0.8, 100)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, int(nk.item())


def time_hbm_copy(device, mbytes=1024, iters=10):
    """Device-to-device copy rate (read + write bytes / time): the practical HBM figure on this
    box next to the 8 TB/s used as `peak`."""
    a = torch.empty(mbytes << 20, dtype=torch.uint8, device=device)
    b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * (mbytes << 20) * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9


def time_fps_shared(pcs, m=4096):
    """Level-1 sampling of the bench batch on the small-footprint sampler the pipelined schedule uses
    (alone on the GPU: slower per iteration than the register-resident kernel, see `fps`)."""
    xyz = pcs[:, :, 0:3].contiguous()
    ops.farthest_point_sample(m, xyz, shared=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.farthest_point_sample(m, xyz, shared=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    return {"ms": round(ms, 3), "us_per_iteration": round(1e3 * ms / (m - 1), 4)}


def time_fps_dense(device, n=131072, m=16384):
    """BASELINE config 5, second half: FPS on a 131072-point dense scene."""
    pts = torch.as_tensor(S.dense_scene(5000, n)[None, :, :3].copy()).to(device)
    ops.farthest_point_sample(64, pts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.farthest_point_sample(m, pts)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    return {"points": n, "npoint": m, "ms": round(ms, 3), "us_per_iteration": round(1e3 * ms / (m - 1), 4)}


def time_frustum(device, store, precision, batch=32, iters=20):
    """BASELINE configs[1] (secondary number, not `value`): the three Frustum PointNet v2 SA/FP
    stacks (cls + seg on 1024-point frustums, box net on 512 masked points), batch 32, as one
    HIP-graph replay per batch."""
    store = PU.ParamStore(seed=9, device=device)     # own weights: the seg net's scope names are the RPN's
    fr = torch.as_tensor(S.frustum_batch(1, batch, 1024)).to(device)
    obj = torch.as_tensor(S.frustum_batch(2, batch, 512, channels=3)).to(device)
    onehot = torch.zeros((batch, 4), device=device); onehot[:, 1] = 1

    def once():
        return (stacks.frustum_cls_stack(fr, params=store, precision=precision),
                stacks.frustum_seg_stack(fr, onehot, params=store, precision=precision),
                stacks.frustum_box_stack(obj, params=store, precision=precision))
    st = torch.cuda.Stream(device=device)
    with torch.cuda.stream(st):
        once(); once()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        keepalive = once()
    with torch.cuda.stream(st):
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    del keepalive
    return {"workload": "Frustum PointNet v2 cls + seg (1024 pts) + box (512 pts) SA/FP stacks, batch %d" % batch,
            "ms_per_batch": round(ms, 3), "frustums_per_s": round(batch / (ms * 1e-3), 1)}


def cpu_baseline(store, seconds_cap=40.0):
    """The CPU oracle (reference CPU path: C++ CPU ops restated + transcribed FPS kernel)
    on ONE 16384-point scene of the same generator, single thread."""
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                           # pragma: no cover
        threadpool_limits = None
    from oracle import oracle as O
    from oracle import stack as ostack
    O.lib()
    params = {k: tuple(t.detach().cpu().numpy() for t in v) for k, v in store.items() if not k.startswith('__')}
    timers = {}
    nscenes, dt = 0, 0.0
    while dt < 10.0 and nscenes < 16:                            # bounded sample: ~10-30 s of CPU work
        pc = S.kitti_batch(nscenes, 1, NPTS)
        t0 = time.perf_counter()
        if threadpool_limits:
            with threadpool_limits(limits=1):
                ostack.rpn_sa_fp_stack(pc, params, gemm='blas', timers=timers)
        else:
            ostack.rpn_sa_fp_stack(pc, params, gemm='blas', timers=timers)
        dt += time.perf_counter() - t0
        nscenes += 1
    dt /= nscenes
    timers = {k: v / nscenes for k, v in timers.items()}
    boxes, scores = S.nms_boxes(seed=7000, N=NPROP, clusters=40)
    t1 = time.perf_counter()
    O.nms_rotate_cpu(boxes, scores, 0.8, 100)
    tn = time.perf_counter() - t1
    b9, s9 = S.nms_boxes(seed=9000, N=9000)
    t1 = time.perf_counter()
    O.nms_rotate_cpu(b9, s9, 0.8, 100)
    tn9 = time.perf_counter() - t1
    # N-core leg: the same single-threaded path farmed over worker processes, one scene
    # stream each; N = this process's CPU share (the box shows the whole machine's CPUs), at most 16
    ncore = None
    try:
        from oracle import cpu_farm
        cores = max(1, min(16, len(os.sched_getaffinity(0))))
        v, n = cpu_farm.farm(params, cores)
        ncore = {"value": round(v, 2), "unit": "scenes/s", "cores": cores,
                 "sample": "%d worker processes x 2 scenes each (stack + NMS 2048->100), 1 BLAS thread per worker" % cores}
    except Exception as ex:                                      # pragma: no cover
        ncore = {"error": repr(ex)}
    return {"value": 1.0 / (dt + tn), "unit": "scenes/s", "cores": 1, "kind": "port", "n_core": ncore,
            "sample": "%d scenes x 16384 pts through the C oracle op by op (+ NumPy fp32 GEMM, 1 BLAS thread) "
                      "+ rotated NMS 2048->100 each; %.2f s per scene" % (nscenes, dt + tn),
            "seconds": {k: round(v, 4) for k, v in timers.items()},
            "nms_ms_2048": tn * 1e3, "nms_ms_9000": tn9 * 1e3}


class PipelinedSchedule:
    """Units of `group` batches as HIP graphs, one per step stream (each with its companion FPS
    stream); a remainder of k % group steps is issued EAGERLY as single-batch units, so exactly k
    steps run for any k and the graphs of the process never depend on K / W (round 1 built extra
    remainder graphs and chose the unit size by the parity of the warm-up count)."""

    def __init__(self, device, nstreams, group, graph, need_remainder, pcs, prop, store, precision):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(nstreams)]
        self.sides = [torch.cuda.Stream(device=device) for _ in range(nstreams)]
        self.group = group
        mk = lambda st, sd, G, gr, pr=precision: Unit(st, sd, pcs, prop, store, pr, G, graph=gr)
        # diagnostic (PRCNN_BENCH_MIXED=1): the second stream runs the fp32 tier beside a bf16 first stream
        mixed = os.environ.get("PRCNN_BENCH_MIXED") == "1"
        self.units = [mk(st, sd, group, graph, ops.PRECISION_FP32 if (mixed and i > 0) else precision)
                      for i, (st, sd) in enumerate(zip(self.streams, self.sides))]
        self.rem = self.units
        if group > 1 and need_remainder:
            self.rem = [mk(st, sd, 1, False) for st, sd in zip(self.streams, self.sides)]
        self.graph_nodes = None

    def run(self, k, ring):
        plan = [self.group] * (k // self.group) + [1] * (k % self.group)
        pos = 0
        stagger = float(os.environ.get("PRCNN_BENCH_STAGGER_MS", "0"))    # probe: offset the second stream's units
        for u, G in enumerate(plan):
            j = u % len(self.streams)
            if stagger > 0 and u == 1:
                time.sleep(stagger * 1e-3)
            unit = self.units[j] if G == self.group else self.rem[j]
            with torch.cuda.stream(self.streams[j]):
                for _, rec in unit.run():
                    ring[pos].copy_(rec)          # the unit rewrites its own records on the next replay
                    pos += 1
        return pos

    def last_outputs(self):
        return self.units[0].outs[-1]


def round_plan(k, nstreams, per_round=12, cap=4):
    """K steps as rounds dealt to the step streams: a list of rounds, each a list of per-stream unit sizes
    (batches; 0 = the stream sits the round out).  Full rounds of per_round steps (equal units: a stream
    that runs a batch more than the others runs it alone) and ONE last round with what is left, its sizes
    within one of each other; every size <= cap.  Exactly K steps for any K."""
    per_round = max(nstreams, min(per_round, cap * nstreams))
    plan = []
    while k > 0:
        kr = min(k, per_round)
        plan.append([kr // nstreams + (1 if i < kr % nstreams else 0) for i in range(nstreams)])
        k -= kr
    return plan


class RoundSchedule:
    """The default pipelined schedule (end of round 2): S step streams and ONE sampling stream.  A round
    runs the feature phases of up to S units (one linear HIP graph per step stream and unit size) beside
    ONE sampling chain over the clouds of the whole next round -- a chain lasts the same for 64 or 240
    clouds -- and ends with every step stream copying its share of the new centroids.  S + 1 lanes
    instead of the 2 S of one forked graph per stream (PipelinedSchedule), so three step streams fit the
    hardware queues: 13.9 k scenes/s against 12.2 k.  Needs >= S + 2 hardware queues
    (bench.py sets 8).  Every batch still gets one complete sampling pass and one complete feature
    phase; a chain always samples S x cap batches, so rounds with smaller units sample some twice."""

    def __init__(self, device, nstreams, steps, warmup, pcs, prop, store, precision, per_round=12, cap=4):
        S = nstreams
        self.S = S
        self.streams = [torch.cuda.Stream(device=device) for _ in range(S)]
        self.sampler = torch.cuda.Stream(device=device)
        self.per_round, self.capmax = per_round, cap
        plans = round_plan(steps, S, per_round, cap) + round_plan(warmup, S, per_round, cap)
        self.cap = max([g for rnd in plans for g in rnd] + [1])
        self.group = self.cap                                    # reported as batches_per_replay
        self.b = pcs.shape[0]
        self.pcs, self.prop, self.store, self.precision = pcs, prop, store, precision
        npoints = [lv[0] for lv in stacks.RPN_SA]
        with torch.cuda.stream(self.sampler):
            xyz_all = pcs[:, :, 0:3].contiguous().repeat(S * self.cap, 1, 1)
            xs, _ = stacks.sample_centroid_chain(xyz_all, npoints)
            self.cent_next = [x.clone() for x in xs]
            ahead_work(pcs, self.cent_next, S * self.cap)        # warm-up of the chain as it is captured
        torch.cuda.synchronize()
        rows = self.cap * self.b
        self.cent_cur = [[c[s * rows:(s + 1) * rows].clone() for c in self.cent_next] for s in range(S)]
        self.units = {}                                          # (stream, size) -> (graph, outs)
        for rnd in plans:
            for s_, g in enumerate(rnd):
                if g and (s_, g) not in self.units:
                    self.units[(s_, g)] = self._capture(s_, g)
        self.g_chain = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_chain, stream=self.sampler):
            ahead_work(pcs, self.cent_next, S * self.cap)
        self.chain_done = torch.cuda.Event()
        self.copied = [torch.cuda.Event() for _ in self.streams]
        self.started = False
        self.last = None

    def _capture(self, s_, g):
        st = self.streams[s_]
        with torch.cuda.stream(st):
            feature_work(self.pcs, self.prop, self.store, self.precision, self.cent_cur[s_], g)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=st):
            outs = feature_work(self.pcs, self.prop, self.store, self.precision, self.cent_cur[s_], g)
        return graph, outs

    def run(self, k, ring):
        rows = self.cap * self.b
        pos = 0
        for rnd in round_plan(k, self.S, self.per_round, self.capmax):
            if self.started:
                for ev in self.copied:                           # cent_next is free once every stream has taken its share
                    self.sampler.wait_event(ev)
            with torch.cuda.stream(self.sampler):
                self.g_chain.replay()
                self.chain_done.record(self.sampler)
            for s_, g in enumerate(rnd):
                st = self.streams[s_]
                with torch.cuda.stream(st):
                    if g:
                        if (s_, g) not in self.units:            # (a size the constructor was not told of: untimed use only)
                            self.units[(s_, g)] = self._capture(s_, g)
                        graph, outs = self.units[(s_, g)]
                        graph.replay()
                        for _, rec in outs:
                            ring[pos].copy_(rec)                 # the unit rewrites its own records on the next replay
                            pos += 1
                        if s_ == 0:
                            self.last = outs[-1]
                    st.wait_event(self.chain_done)
                    for dst, src in zip(self.cent_cur[s_], self.cent_next):
                        dst.copy_(src[s_ * rows:(s_ + 1) * rows])
                    self.copied[s_].record(st)
            self.started = True
        return pos

    def last_outputs(self):
        return self.last


class SimpleSchedule:
    """One batch per unit, FPS inside the batch's own step (--no-pipeline / --no-graph / --geo-stream)."""

    def __init__(self, device, nstreams, graph, geo_stream, pcs, prop, store, precision):
        self.streams = [torch.cuda.Stream(device=device) for _ in range(nstreams)]
        self.geos = [torch.cuda.Stream(device=device) if geo_stream else None for _ in self.streams]
        self.call = lambda geo: step(pcs, prop, store, precision, geo)
        self.graphs, self.outs = [], []
        for st, geo in zip(self.streams, self.geos):
            with torch.cuda.stream(st):
                o = self.call(geo)                                  # lazy init + allocator warm-up
            torch.cuda.synchronize()
            g = None
            if graph:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    o = self.call(geo)
            self.graphs.append(g); self.outs.append(o)

    def run(self, k, ring):
        for i in range(k):
            j = i % len(self.streams)
            with torch.cuda.stream(self.streams[j]):
                if self.graphs[j] is not None:
                    self.graphs[j].replay()
                else:
                    self.outs[j] = self.call(self.geos[j])
                ring[i].copy_(self.outs[j][1])
        return k

    def last_outputs(self):
        return self.outs[0]


def timed_region(sched, steps, warmup, ring, device, world):
    """W untimed steps, then EXACTLY K timed steps between fences (synchronize + barrier on both
    sides); the one all-gather of the region's records is issued inside it, at its closing fence.
    Returns (elapsed max over ranks, host issue time, gathered records of the timed steps)."""
    def fence(count):
        torch.cuda.synchronize()
        gathered = ring[:count].reshape(-1, pdist.RECORD_LEN)
        if dist.is_initialized() and count:
            gathered = pdist.all_gather_detections(gathered)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        return gathered

    fence(sched.run(warmup, ring))
    t0 = time.perf_counter()
    count = sched.run(steps, ring)
    t_issue = time.perf_counter() - t0
    gathered = fence(count)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed, t_issue, gathered


def time_single_scene(device, store, precision, pcs, iters=10):
    """BASELINE configs[0]'s shape on the GPU: ONE 16384-point scene through the SA+FP stack, nothing
    else in flight -- the latency of a scene, not throughput (the FPS chain, 5439 strictly serial
    iterations, is on the critical path here).  As one HIP-graph replay, the FPS chain inside."""
    pc = pcs[:1].contiguous()
    st = torch.cuda.Stream(device=device)
    with torch.cuda.stream(st):
        stacks.rpn_sa_fp_stack(pc, params=store, precision=precision)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        keep = stacks.rpn_sa_fp_stack(pc, params=store, precision=precision)
    with torch.cuda.stream(st):
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            g.replay()
        e1.record()
    torch.cuda.synchronize()
    del keep
    return round(e0.elapsed_time(e1) / iters, 3)


def load_profile(name):
    """A committed rocprofv3 summary under profiles/ (newest round first), or {}."""
    for rnd in ("r02", "r01"):
        path = os.path.join(ROOT, "profiles", "%s_%s" % (rnd, name))
        if os.path.exists(path):
            try:
                d = json.load(open(path))
                d["__file__"] = "profiles/%s_%s" % (rnd, name)
                return d
            except Exception:
                pass
    return {}


# (kernel family, shape) -> entry of profiles/*_pmc_hot_kernels.json (tools/pmc_probe.py launches these shapes)
PMC_KEYS = {
    ("mlp_layer", "rows65536 K608 N256 pool1"): "mlp_kernel<2, 2, 2, 2, false, false, 32, false> grid_threads=262144",
    ("mlp_layer", "rows262144 K260 N128 pool1"): "mlp_kernel<2, 2, 2, 2, false, false, 32, false> grid_threads=524288",
    ("mlp_layer", "rows524288 K96 N128 pool32"): "mlp_kernel<2, 2, 2, 2, false, false, 16, false> grid_threads=1048576",
    ("sa_chain", "rows2097152 K4 N32-32-64 pool32"): "sa_chain_kernel<4, 32, 32, 64, 32, 32>",
    ("sa_chain2", "rows524288 N64-96-128 pool32"): "sa_chain2_kernel<96, 6>",
    ("mlp_layer", "rows4096 K512 N512 pool1"): "mlp_kernel<4, 1, 1, 2, false, false, 32, false> grid_threads=65536",
    ("three_interpolate", None): "three_interpolate_kernel<4>",
    ("group_point", None): "group_point_kernel<4>",
    ("three_nn_interpolate", None): "ns_search_kernel<4, true>",
}
HBM_KERNELS = ("group_point", "three_interpolate", "three_nn_interpolate", "gather_point", "sa_group_add_layer")


def pmc_entry(pmc, kernel, shape):
    key = PMC_KEYS.get((kernel, shape)) or PMC_KEYS.get((kernel, None))
    e = pmc.get(key) if key else None
    return e if e and "hbm_bytes_corrected" in e else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32")
    # concurrent units in flight (round schedule: + 1 sampling stream; forked-graph schedule: each step stream has a
    # companion stream for its chain, and 2+2 streams = HIP's 4 hardware queues -- 3+3 aliases queues and loses 20 %)
    ap.add_argument("--streams", type=int, default=0,
                    help="step streams (0 = 3 for the round schedule, 2 for the forked-graph schedule)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed region and the instrumented pass (no NMS / dense FPS / Frustum / no-dedup legs)")
    ap.add_argument("--geo-stream", action="store_true",
                    help="run each step's FPS chain on a companion stream (helps only single-batch latency)")
    ap.add_argument("--no-graph", action="store_true", help="issue every step eagerly instead of replaying HIP graphs")
    ap.add_argument("--fps-group", type=int, default=0,
                    help="batches per graph replay: one FPS chain samples this many upcoming batches at once "
                         "(0 = automatic: the size in 4, 5, 6, 3, 8 that deals the K steps evenly over the step "
                         "streams, so no stream idles while another finishes a last unit; else 4 + eager remainder)")
    ap.add_argument("--no-dedup", action="store_true",
                    help="evaluate every padded ball slot (pointnet_util.SA_DEDUP = False): the cost when every ball is "
                         "full; by default the copies of a ball's first hit that query_ball_point pads with are skipped")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="sample each batch's centroids inside its own step instead of one step ahead")
    ap.add_argument("--dump-records", default=None,
                    help="rank 0 saves the (all-gathered) detection records of the timed steps to this .npy file")
    args = ap.parse_args()
    if os.environ.get("PRCNN_BENCH_WATCHDOG"):        # debugging aid: dump all stacks and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PRCNN_BENCH_WATCHDOG"]), exit=True)
    if os.environ.get("PRCNN_SEGV_BACKTRACE"):        # debugging aid: native backtrace on SIGSEGV / SIGABRT
        import ctypes
        ctypes.CDLL(os.path.join(ROOT, "tools", "libsegv_bt.so")).segv_bt_install()

    PU.SA_DEDUP = not args.no_dedup
    # Hardware queues.  The schedule keeps four streams busy (two step streams + the FPS branch of each
    # unit's graph) and its throughput depends on how HIP deals its hardware queues to the process's
    # live streams: with the default of 4 queues a run WITHOUT a process group is at full speed, but
    # one more live stream -- RCCL's, or any idle stream created earlier -- costs 24 % (measured:
    # 10.8k -> 8.3k scenes/s); with 8 queues the run WITH a one-rank RCCL group is back at 10.9k (and the
    # run without a group drops to 7.6k).  DESIGN.md 5 has the table.  So: 8 queues when a process group
    # will be formed, the default otherwise (PRCNN_HW_QUEUES overrides); the variable is read when HIP
    # initialises (first HIP call, below).
    # The round schedule (default; RoundSchedule) runs S + 1 = 4 lanes of linear graphs and wants a queue per lane
    # plus the null stream's: 5, 6 and 8 queues measure alike (13.9 k scenes/s), 4 lose a fifth (10.8 k).
    grouped = int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("PRCNN_DIST_FORCE") == "1"
    pipelined = not (args.no_graph or args.no_pipeline or args.geo_stream)
    rounds = (pipelined and os.environ.get("PRCNN_BENCH_ROUNDS", "1") != "0" and args.fps_group <= 0
              and os.environ.get("PRCNN_BENCH_PROBE_NO_FPS") not in ("1", "6") and os.environ.get("PRCNN_BENCH_TAP") != "1")
    if os.environ.get("PRCNN_HW_QUEUES"):                       # explicit choice for this bench
        os.environ["GPU_MAX_HW_QUEUES"] = os.environ["PRCNN_HW_QUEUES"]
    elif (grouped or rounds) and os.environ.get("GPU_MAX_HW_QUEUES", "4") == "4":   # unset, or exported at HIP's own default
        os.environ["GPU_MAX_HW_QUEUES"] = "8"
    rank, world, local = pdist.init_from_env("cuda")
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    precision = ops.PRECISION_FP32 if args.precision == "f32" else ops.PRECISION_BF16
    store = PU.ParamStore(seed=7, device=device)
    # PRCNN_BENCH_INPUT_RANK: run rank r's scenes in a one-process run (tests compare them with the N > 1 job's)
    pcs, prop = make_inputs(int(os.environ.get("PRCNN_BENCH_INPUT_RANK", rank)), device, args.batch)
    nstreams = args.streams if args.streams > 0 else (3 if rounds else 2)
    group = max(1, args.fps_group) if pipelined else 1
    if pipelined and args.fps_group <= 0:
        even = [g for g in (4, 5, 6, 3, 8) if args.steps % (g * nstreams) == 0]
        group = even[0] if even else 4
    ring = torch.zeros((max(args.steps, args.warmup, 1), args.batch, pdist.RECORD_LEN), dtype=torch.float32, device=device)

    def make_schedule():
        if rounds:
            return RoundSchedule(device, nstreams, args.steps, args.warmup, pcs, prop, store, precision,
                                 per_round=int(os.environ.get("PRCNN_BENCH_ROUND_STEPS", "12")),
                                 cap=int(os.environ.get("PRCNN_BENCH_ROUND_CAP", "4")))
        if pipelined:
            return PipelinedSchedule(device, nstreams, group, True, bool(args.steps % group or args.warmup % group),
                                     pcs, prop, store, precision)
        return SimpleSchedule(device, nstreams, not args.no_graph, args.geo_stream, pcs, prop, store, precision)

    # probe: streams created (and used once) before the schedule's, to shift the stream -> hardware queue
    # assignment of the streams that matter (HIP hands its 4 hardware queues out in creation order)
    pad_streams = [torch.cuda.Stream(device=device) for _ in range(int(os.environ.get("PRCNN_BENCH_PAD_STREAMS", "0")))]
    for ps in pad_streams:
        with torch.cuda.stream(ps):
            torch.zeros(1, device=device)
    torch.cuda.synchronize()
    if os.environ.get("PRCNN_BENCH_PAD_DROP") == "1":
        del pad_streams, ps
        import gc
        gc.collect()
    tap = os.environ.get("PRCNN_BENCH_TAP") == "1"
    if tap:
        PU.DEBUG_TAP = []
    sched = make_schedule()
    taps_graph = list(PU.DEBUG_TAP) if tap else None
    elapsed, t_issue, gathered = timed_region(sched, args.steps, args.warmup, ring, device, world)

    # The replayed graphs against an eager, un-pipelined step on the same inputs (every step of this
    # synthetic run sees the same batch): same features, same records, bit for bit.
    feats_replay, rec_replay = sched.last_outputs()
    if tap:
        PU.DEBUG_TAP = []
    with torch.cuda.stream(sched.streams[0]):
        feats_eager, rec_eager = step(pcs, prop, store, precision)
    torch.cuda.synchronize()
    if tap:
        # unit 0 was built first: its taps are [eager warm-up x group, captured x group]; the captured tensors of its
        # last batch hold what the last replay computed
        g0 = sched.group
        cap = taps_graph[2 * g0 - 1]
        for (sc, y2g, xg, y2b, y2a), (_, y2e, xe, _b, _a) in zip([cap], PU.DEBUG_TAP[-1:]):
            print("bench tap %s: copy of y2 taken before the search equals final y2: %s (%d differ); copy after: %s (%d differ)"
                  % (sc, bool(torch.equal(y2b, y2g)), int((y2b != y2g).sum().item()), bool(torch.equal(y2a, y2g)),
                     int((y2a != y2g).sum().item())), file=sys.stderr)
            print("bench tap %s: y2 equal %s (%d differ), interpolated equal %s (%d differ)"
                  % (sc, bool(torch.equal(y2g, y2e)), int((y2g != y2e).sum().item()), bool(torch.equal(xg, xe)),
                     int((xg != xe).sum().item())), file=sys.stderr)
            d = (xg != xe).reshape(-1, xe.shape[-1])
            for r in d.any(1).nonzero().flatten()[:6].tolist():
                cols = d[r].nonzero().flatten().tolist()
                print("bench tap   row %d (cloud %d, point %d) cols %s graph %s eager %s" % (
                    r, r // xe.shape[1], r % xe.shape[1], cols, xg.reshape(-1, xe.shape[-1])[r, cols[:4]].tolist(),
                    xe.reshape(-1, xe.shape[-1])[r, cols[:4]].tolist()), file=sys.stderr)
    own = ring[:args.steps]
    replay_ok = bool(torch.equal(feats_replay, feats_eager) and torch.equal(rec_replay, rec_eager)
                     and all(torch.equal(own[i], rec_eager) for i in range(own.shape[0])))
    replay_diff = None
    if not replay_ok:
        nf = int((feats_replay != feats_eager).sum().item())
        rows = (feats_replay != feats_eager).reshape(-1, feats_eager.shape[-1]).any(1).nonzero().flatten()[:8].tolist()
        fr, fe = feats_replay.reshape(-1, feats_eager.shape[-1]), feats_eager.reshape(-1, feats_eager.shape[-1])
        for r in rows[:4]:
            cols = (fr[r] != fe[r]).nonzero().flatten().tolist()
            lo, hi = max(0, r - 300), min(fe.shape[0], r + 300)
            match = (fe[lo:hi] == fr[r]).all(1).nonzero().flatten().tolist()
            print("bench: row %d: %d columns differ (first %s), replay row equals eager rows %s; replay %s eager %s"
                  % (r, len(cols), cols[:6], [lo + q for q in match[:4]], fr[r, cols[:3]].tolist(), fe[r, cols[:3]].tolist()),
                  file=sys.stderr)
        maxd = float((feats_replay - feats_eager).abs().max().item()) if nf else 0.0
        recs_ok = bool(torch.equal(rec_replay, rec_eager)) and all(bool(torch.equal(own[i], rec_eager)) for i in range(own.shape[0]))
        msg = ("bench: graph replays differ from the eager step: %d feature elements (max |diff| %.3g, rows %s), records equal: %s"
               % (nf, maxd, rows, recs_ok))
        # fp32 tier: bit-identity is required.  bf16 tier: with TWO bf16 streams in flight a handful of elements of the
        # last FP level's interpolation differ between runs (never with one stream, never beside an fp32 stream, never
        # in fp32: DESIGN.md 5, open finding); the run is kept if the difference is inside the tier's tolerance and
        # the line reports it.
        if args.precision != "bf16" or not recs_ok or maxd > 1e-2 * float(feats_eager.abs().max().item()):
            raise SystemExit(msg)
        print(msg + " -- inside the bf16 tier's tolerance, reported as config.replay_diff", file=sys.stderr)
        replay_diff = {"elements": nf, "of": feats_eager.numel(), "max_abs": round(maxd, 6)}
    if args.dump_records and rank == 0:
        np.save(args.dump_records, gathered.detach().cpu().numpy())

    # Instrumented pass (eager, ONE stream, HIP events around every launch: no kernel overlaps another):
    # per-kernel durations for the roofline numbers.  Not part of `value`.
    timer = ops.KernelTimer()
    prev = ops.set_timer(timer)
    try:
        with torch.cuda.stream(sched.streams[0]):
            for _ in range(min(args.steps, 4)):
                step(pcs, prop, store, precision)
    finally:
        ops.set_timer(prev)
    torch.cuda.synchronize()
    n_streams_used = len(sched.streams)
    if rounds:
        group = sched.group
        round_sizes = round_plan(args.steps, nstreams, sched.per_round, sched.capmax)
    hip_graph = not args.no_graph
    del sched, feats_replay, rec_replay, feats_eager, rec_eager

    if rank == 0:
        scenes = args.batch * args.steps * world
        peak_mfma = roofline.FP32_MFMA_PEAK_TFLOPS if precision == 0 else roofline.BF16_MFMA_PEAK_TFLOPS
        pmc = load_profile("pmc_hot_kernels.json")
        ksum = timer.summary()
        kernels = {}
        for name, d in ksum.items():
            e = {"launches": d["launches"], "ms_total": round(d["ms"], 3),
                 "avg_us": round(1e3 * d["ms"] / d["launches"], 2)}
            if d.get("stalled"):
                e["host_stalled_samples"] = d["stalled"]      # replaced by their shape's median (KernelTimer.summary)
            if "flops" in d:
                e.update(bound="mfma", achieved=round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 3), peak=peak_mfma, unit="TFLOP/s")
                e["frac"] = round(e["achieved"] / e["peak"], 4)
            elif name in HBM_KERNELS:
                # `achieved` = ALGORITHMIC bytes / time (SURVEY.md 8d's formula counts the gathered reads
                # that L2 / Infinity Cache serve, so it may exceed the HBM peak); `frac` is therefore taken
                # from measured HBM traffic (PMC) where a committed pass covers the kernel, and is absent
                # otherwise when the algorithmic rate is above the peak.
                alg = d["bytes"] / (d["ms"] * 1e-3) / 1e9
                e.update(bound="hbm", achieved=round(alg, 3), peak=roofline.HBM_PEAK_GBS, unit="GB/s")
                if alg <= roofline.HBM_PEAK_GBS:       # above the peak it is a cache-served rate, not a fraction of HBM
                    e["frac_algorithmic"] = round(alg / roofline.HBM_PEAK_GBS, 4)
                pe = pmc_entry(pmc, name, None)
                if pe and pe.get("avg_us"):
                    e["frac"] = round(min(pe["hbm_bytes_corrected"] / (pe["avg_us"] * 1e-6) / 1e9 / roofline.HBM_PEAK_GBS, 1.0), 4)
                    e["frac_source"] = "PMC traffic (%s, largest RPN shape): corrected %d B / %.1f us" % (
                        pmc.get("__file__"), int(pe["hbm_bytes_corrected"]), pe["avg_us"])
                elif alg <= roofline.HBM_PEAK_GBS:
                    e["frac"] = e["frac_algorithmic"]
            if "tests" in d:
                e["gtests_per_s"] = round(d["tests"] / (d["ms"] * 1e-3) / 1e9, 2)
            kernels[name] = e
        # dominant roofline-bound kernel = the single (kernel, shape) with the most time in the
        # step among the HBM- / MFMA-bound ones
        best = None
        for (name, _), d in timer.summary(by_shape=True).items():
            if "flops" in d:
                cand = dict(kernel=name, shape=d.get("shape"), bound="mfma", unit="TFLOP/s", peak=peak_mfma,
                            achieved=d["flops"] / (d["ms"] * 1e-3) / 1e12, algorithmic=d["flops"] // d["launches"])
            elif name in ("group_point", "three_interpolate", "three_nn_interpolate", "sa_group_add_layer"):
                cand = dict(kernel=name, shape=d.get("shape"), bound="hbm", unit="GB/s", peak=roofline.HBM_PEAK_GBS,
                            achieved=d["bytes"] / (d["ms"] * 1e-3) / 1e9, algorithmic=d["bytes"] // d["launches"])
                if "tests" in d:
                    cand["gtests_per_s"] = round(d["tests"] / (d["ms"] * 1e-3) / 1e9, 2)
            else:
                continue
            cand.update(ms=d["ms"], launches=d["launches"], avg_launch_us=round(1e3 * d["ms"] / d["launches"], 2))
            if best is None or cand["ms"] > best["ms"]:
                best = cand
        roof = None
        if best:
            traffic, tsrc = None, None
            pe = pmc_entry(pmc, best["kernel"], best["shape"]) if (precision == 0 and args.batch == BATCH) else None
            if pe:
                traffic = pe["hbm_bytes_corrected"]
                tsrc = ("%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_probe.py on this kernel "
                        "and shape, read side x2 (gfx950); raw %d B" % (pmc.get("__file__"), int(pe["hbm_bytes_raw"])))
            frac = best["achieved"] / best["peak"]
            roof = {"kernel": best["kernel"], "shape": best["shape"], "bound": best["bound"],
                    "achieved": round(best["achieved"], 3), "peak": best["peak"], "unit": best["unit"],
                    "frac": round(frac, 4), "traffic": traffic,
                    "traffic_source": tsrc, "algorithmic_per_launch": best["algorithmic"],
                    "avg_launch_us": best["avg_launch_us"], "launches": best["launches"]}
            if best["bound"] == "hbm":
                # an HBM-bound kernel's fraction comes from the measured traffic, never from gathered
                # (cache-served) algorithmic bytes: a number above 1 is not a roofline fraction
                if roof["frac"] <= 1.0:
                    roof["frac_algorithmic"] = roof["frac"]
                roof["frac"] = (round(min(traffic / (best["avg_launch_us"] * 1e-6) / 1e9 / best["peak"], 1.0), 4)
                                if traffic else (roof["frac"] if roof["frac"] <= 1.0 else None))
                if "gtests_per_s" in best:
                    roof["gtests_per_s"] = best["gtests_per_s"]
        fps = None
        if "fps" in ksum:
            d = ksum["fps"]
            # per launch all clouds of the batch run concurrently (one workgroup each)
            fps = {"bound": "latency", "us_per_iteration": round(1e3 * d["ms"] / (d["iters"]), 4),
                   "ms_total": round(d["ms"], 3), "launches": d["launches"],
                   "gdist_evals_per_s": round(d["dist_evals"] / (d["ms"] * 1e-3) / 1e9, 2)}
            l1 = [dd for (nm, _), dd in timer.summary(by_shape=True).items()
                  if nm == "fps" and dd.get("iters", 0) // max(dd["launches"], 1) == 4095]
            if l1:
                us = 1e3 * l1[0]["ms"] / l1[0]["iters"]
                # one min-distance read + write per point per iteration, had the array lived in
                # LDS; here it lives in VGPRs, so this is the rate the register-resident design replaces
                fps["level1_16384_to_4096"] = {"us_per_iteration": round(us, 4),
                                               "equivalent_lds_gbytes_per_s_per_cloud": round(16384 * 8 / (us * 1e-6) / 1e9, 1)}
            pf = load_profile("pmc_fps.json")
            if pf:
                fps["pmc"] = {k: v for k, v in pf.items() if not k.startswith("__")}
                fps["pmc_source"] = pf["__file__"]
        out = {
            "metric": "SA/FP-stack scenes/sec (16k pts)", "value": round(scenes / elapsed, 2), "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "PointRCNN RPN SA+FP stack, batch %d x %d pts per GPU, fused group->shared-MLP"
                                   "->max-pool (MFMA), + rotated NMS %d->100 per scene%s" %
                                   (args.batch, NPTS, NPROP, " + RCCL all-gather of detections" if world > 1 else ""),
                       "batch_per_gpu": args.batch, "points": NPTS, "streams": n_streams_used,
                       "hip_graph": hip_graph, "host_issue_ms_per_step": round(1e3 * t_issue / args.steps, 3),
                       "geometry_streams": bool(args.geo_stream), "fps_one_step_ahead": bool(pipelined),
                       "fps_level1_sampler": (("shared (L2-resident buckets, 4 waves per cloud)"
                                               if os.environ.get("PRCNN_BENCH_FPS_SHARED", "1") != "0"
                                               else "register-resident (one CU per cloud)") if pipelined else "register-resident (one CU per cloud)"),
                       "duplicate_free_groups": bool(PU.SA_DEDUP), "ball_fill": ball_fill(pcs),
                       "batches_per_replay": group, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "eager_remainder_steps": (args.steps % group) if (pipelined and not rounds) else 0,
                       "schedule": ("rounds: %d step streams (one linear graph per unit) + 1 sampling stream (one chain per round); "
                                    "unit sizes per round %s" % (nstreams, round_sizes) if rounds else
                                    ("one forked graph per step stream" if pipelined else "one batch per unit")),
                       "replay_equals_eager": replay_ok, "replay_diff": replay_diff,
                       "kernel_timing": "HIP events around every launch in an eager, single-stream instrumented "
                                        "pass of the same steps after the timed region",
                       "parallelism": "scene-sharded x%d" % world,
                       "collective": ("one %s all-gather of all detection records per timed region"
                                      % ("RCCL" if dist.get_backend() == "nccl" else dist.get_backend() + " (rehearsal)")
                                      if dist.is_initialized() else None)},
            "roofline": roof, "kernels": kernels, "fps": fps,
        }
        if not args.no_extras:
            if fps is not None:
                fps["level1_shared_sampler"] = time_fps_shared(pcs)
            nms_ms, nms_keep = time_nms(device)
            hbm_copy = time_hbm_copy(device)
            out.update({
                "nms": {"ms_9000_to_100": round(nms_ms, 4), "kept": nms_keep, "pairs": 9000 * 8999 // 2,
                        "metric": "rotated-NMS ms @9k boxes"},
                "fps_dense": time_fps_dense(device),
                "frustum": time_frustum(device, store, precision),
                "single_scene_latency_ms": time_single_scene(device, store, precision, pcs),
                "hbm_copy_gbps": {"value": round(hbm_copy, 1), "what": "torch device-to-device copy of 1 GiB, read+write bytes / time"},
            })
            if world == 1 and PU.SA_DEDUP:
                # the same timed region with every padded ball slot evaluated: what the number is when
                # every ball is full (the headline skips ~90 % of the grouped rows on these scenes, exactly)
                PU.SA_DEDUP = False
                try:
                    torch.cuda.empty_cache()
                    s2 = make_schedule()
                    e2, _, _ = timed_region(s2, args.steps, args.warmup, ring, device, world)
                    out["value_no_dedup"] = round(scenes / e2, 2)
                    del s2
                finally:
                    PU.SA_DEDUP = True
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(store)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
