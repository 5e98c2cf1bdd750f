#!/usr/bin/env python3
"""Benchmark of the PointRCNN point-cloud hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N>1: launched by torch.distributed.run, one rank per GPU over RCCL.)

Workload (BASELINE.json configs[2], the config the metric "SA/FP-stack scenes/sec
(16k pts)" is quoted on): the PointRCNN RPN SA+FP stack on a batch of 16 synthetic
KITTI-shaped scenes x 16384 points per GPU, fused group->shared-MLP->max-pool on the
matrix cores, followed by the rotated BEV NMS of each scene's 2048 proposals (top 100,
IoU 0.8: the reference's val/test setting) and -- for N>1 -- ONE RCCL all-gather of all
per-scene detection records produced in the timed region, issued at its closing fence.  One "step" = one such batch per rank (weak scaling: every
rank processes its own 16 scenes per step).  Inputs are resident in HBM before the timed
region.  Steps are issued as HIP-graph replays of `--fps-group` (default 4) batches each,
alternating over two streams (a remainder of K mod 4 steps runs as single-batch replays, so
exactly K steps are timed).  Inside a replay the strictly serial farthest-point-sampling
chain runs ONCE over the clouds of that stream's NEXT four batches (64 workgroups, 6.3 ms
whatever the number of clouds) beside the full-chip feature phases of the current four
(software pipelining: every batch still gets one complete FPS pass and one complete feature
phase, see pipelined_step; --no-pipeline samples inside the batch's own step, --fps-group 1
pipelines batch by batch).

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant roofline-bound
kernel, durations measured with HIP events inside the timed region), `kernels` (every
kernel family), `fps` (the latency-bound sampler: us/iteration), `nms` (ms @ 9000 boxes),
`cpu_baseline` (the CPU oracle = reference CPU path, 1 core, bounded sample).
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


def step(pcs, prop, store, precision, geo=None, centroids=None, geometry=None, rec=None):
    """SA+FP stack on the batch, per-scene rotated NMS of its proposals, detection records."""
    feats, _ = stacks.rpn_sa_fp_stack(pcs, params=store, precision=precision, geometry_stream=geo,
                                      centroids=centroids, geometry=geometry)
    return feats, (detections(prop) if rec is None else rec)


def ahead_work(pcs_next, prop, cent_next, geo_next, group):
    """What runs AHEAD of (and beside) the feature phases: ONE strictly serial FPS chain over the
    clouds of the next `group` batches into `cent_next`; with `geo_next` also everything else
    that depends on coordinates only (ball queries of all SA scales, three_nn + weights of all FP
    levels: stacks.rpn_geometry) into `geo_next`, and the rotated NMS + detection records of the
    CURRENT batches (independent of the stack).  Returns the records ([None]*group without
    `geo_next`)."""
    xyz_next = pcs_next[:, :, 0:3].contiguous()
    if group > 1:
        xyz_next = xyz_next.repeat(group, 1, 1)              # synthetic input: the same batch again
    xs, _ = stacks.sample_centroid_chain(xyz_next, [lv[0] for lv in stacks.RPN_SA])
    for dst, x in zip(cent_next, xs):
        dst.copy_(x)
    if geo_next is None:
        return [None] * group
    g_next = stacks.rpn_geometry(xyz_next, xs)
    for dst, src in zip(stacks.geometry_tensors(geo_next), stacks.geometry_tensors(g_next)):
        dst.copy_(src)
    return [detections(prop) for _ in range(group)]


def feature_work(pcs_cur, prop, store, precision, cent_cur, geo_cur, group, recs):
    """The feature phases (grouping, MLPs, interpolation; NMS + records unless `recs` has them) of
    the current `group` batches with centroids (and geometry) prepared one replay earlier."""
    b = pcs_cur.shape[0]
    return [step(pcs_cur, prop, store, precision, centroids=[c[g * b:(g + 1) * b] for c in cent_cur],
                 geometry=None if geo_cur is None else stacks.geometry_slice(geo_cur, g * b, (g + 1) * b),
                 rec=recs[g])
            for g in range(group)]


def pipelined_step(pcs_cur, pcs_next, prop, store, precision, cent_cur, cent_next, side, group=1,
                   geo_cur=None, geo_next=None):
    """`group` steps of the software-pipelined schedule as ONE unit of work forked over two
    streams: feature_work of the current `group` batches on the current stream, ahead_work for
    the next `group` batches on `side`.  Every batch still gets one complete FPS pass and one
    complete feature phase; only their pairing is shifted, so the latency-bound sampling (one
    chain costs the same whatever the number of clouds, 16 CUs per batch) never gates full-chip
    work, and one chain's latency is spread over `group` batches.  Returns one (feats, rec) per
    batch."""
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        recs = ahead_work(pcs_next, prop, cent_next, geo_next, group)
    outs = feature_work(pcs_cur, prop, store, precision, cent_cur, geo_cur, group, recs)
    cur.wait_stream(side)
    return outs


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
    return {"value": 1.0 / (dt + tn), "unit": "scenes/s", "cores": 1, "kind": "port",
            "sample": "%d scenes x 16384 pts through the C oracle op by op (+ NumPy fp32 GEMM, 1 BLAS thread) "
                      "+ rotated NMS 2048->100 each; %.2f s per scene" % (nscenes, dt + tn),
            "seconds": {k: round(v, 4) for k, v in timers.items()},
            "nms_ms_2048": tn * 1e3, "nms_ms_9000": tn9 * 1e3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32")
    # concurrent batches in flight; each has a companion stream for the next batch's FPS chain,
    # and 2+2 streams = HIP's 4 hardware queues (measured: 3+3 aliases queues and loses 20 %)
    ap.add_argument("--streams", type=int, default=2)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--geo-stream", action="store_true",
                    help="run each step's FPS chain on a companion stream (helps only single-batch latency)")
    ap.add_argument("--no-graph", action="store_true", help="issue every step eagerly instead of replaying HIP graphs")
    ap.add_argument("--fps-group", type=int, default=4,
                    help="batches per graph replay: one FPS chain samples this many upcoming batches at once")
    ap.add_argument("--geo-ahead", action="store_true",
                    help="pipelined schedule: answer ball queries / three_nn (+ NMS) on the FPS side stream one "
                         "replay ahead instead of in line with the feature phase (measured: 6510 vs 6757 scenes/s "
                         "in line -- the chip is bound by total work / power, not by stream order; kept as an option)")
    ap.add_argument("--no-dedup", action="store_true",
                    help="evaluate every padded ball slot (pointnet_util.SA_DEDUP = False): the cost when every ball is "
                         "full; by default the copies of a ball's first hit that query_ball_point pads with are skipped")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="sample each batch's centroids inside its own step instead of one step ahead")
    args = ap.parse_args()
    if os.environ.get("PRCNN_BENCH_WATCHDOG"):        # debugging aid: dump all stacks and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["PRCNN_BENCH_WATCHDOG"]), exit=True)

    PU.SA_DEDUP = not args.no_dedup
    rank, world, local = pdist.init_from_env("cuda")
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    precision = ops.PRECISION_FP32 if args.precision == "f32" else ops.PRECISION_BF16
    store = PU.ParamStore(seed=7, device=device)
    pcs, prop = make_inputs(rank, device, args.batch)
    streams = [torch.cuda.Stream(device=device) for _ in range(max(1, args.streams))]
    # each step stream has a companion stream for its farthest-point-sampling chain
    geos = [torch.cuda.Stream(device=device) if args.geo_stream else None for _ in streams]

    # One HIP graph of the whole step per stream (the C ABI allocates nothing and never
    # synchronises, so it captures as is); the all-gather stays outside the graph.
    graphs, outs = [], []
    two_graph = False
    pipelined = not (args.no_graph or args.no_pipeline or args.geo_stream)
    # HIP-graph replay on this ROCm is fragile: units of 6 batches (~500 kernel nodes per graph) and,
    # measured later, of 3 batches segfault inside hipGraphLaunch, 1 / 2 / 4 / 5 run (4: thousands of
    # replays over the round).  Only the measured-safe sizes are offered.
    group = {1: 1, 2: 2, 3: 2, 4: 4}.get(max(1, args.fps_group), 5) if pipelined else 1
    # The live graphs of the process must also stay under ~1600 nodes in total (see below).  With the
    # duplicate-free launches a batch is ~85 nodes, so units of 4 fit (4 graphs x ~330) but units of 5,
    # or units of 4 plus the single-batch remainder graphs, do not: when K or W is not a multiple of
    # the unit, fall back to a unit that divides both, else to units of 2 + single batches.
    if pipelined:
        group = min(group, 4)
        if args.steps % group or args.warmup % group:
            fits = [g for g in (4, 2) if g <= group and args.steps % g == 0 and args.warmup % g == 0]
            group = fits[0] if fits else min(group, 2)
        # the same budget over the step streams (3 streams x units of 4 = ~2000 nodes: crashed)
        both_kinds = lambda g: bool(args.steps % g or args.warmup % g)
        while group > 1 and max(1, args.streams) * 2 * (85 * group + 30 + (115 if both_kinds(group) else 0)) > 1500:
            group = 2 if group > 2 else 1
    if pipelined:
        # two graphs per stream and unit size (ping-pong over two centroid sets): replay r reads
        # set r%2, which replay r-1 of the same kind on the same stream filled, and fills set
        # (r+1)%2.  Units of `group` batches carry the bulk; single-batch units serve the
        # remainder when the step count is not a multiple of `group`.
        sides = [torch.cuda.Stream(device=device) for _ in streams]
        # --geo-ahead: the ahead work (FPS chain + geometry + NMS) and the feature work
        # are two graphs per replay unit, launched on the side and the main stream with stream-level
        # ordering between replays.  The side graph fills set B (centroids + geometry of the next
        # batches, one flat buffer), the main graph reads set A, and ONE device copy B -> A closes
        # the replay (~100 MB per 4 batches: 1 % of a replay), so a unit needs one pair of graphs, not
        # a ping-pong of two.  This ROCm segfaults inside hipGraphLaunch once the live graphs of
        # the process hold more than ~1600 nodes in total (measured: 1500 run, 1660 and 1740
        # crash, whatever their shape); this layout holds ~850.
        two_graph = args.geo_ahead
        for st, side in zip(streams, sides):
            kinds = {}
            for G in sorted({group, 1} if (args.steps % group or args.warmup % group) else {group}):
                with torch.cuda.stream(st):
                    xyz_rep = pcs[:, :, 0:3].contiguous().repeat(G, 1, 1)
                    xs, _ = stacks.sample_centroid_chain(xyz_rep, [lv[0] for lv in stacks.RPN_SA])
                    if two_graph:
                        g0 = stacks.rpn_geometry(xyz_rep, xs)
                        tens = list(xs) + stacks.geometry_tensors(g0)
                        flat_a, va = stacks.flat_copy_set(tens)
                        flat_b, vb = stacks.flat_copy_set(tens)
                        cents = [va[:len(xs)], vb[:len(xs)]]
                        gsets = [stacks.geometry_from_tensors(g0, va[len(xs):]),
                                 stacks.geometry_from_tensors(g0, vb[len(xs):])]
                    else:
                        cents = [[x.clone() for x in xs], [x.clone() for x in xs]]
                        gsets = [None, None]
                    pipelined_step(pcs, pcs, prop, store, precision, cents[0], cents[1], side, G,
                                   gsets[0], gsets[1])                                              # warm-up
                torch.cuda.synchronize()
                if two_graph:
                    g_side, g_main = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g_side, stream=side):
                        recs = ahead_work(pcs, prop, cents[1], gsets[1], G)
                    with torch.cuda.graph(g_main, stream=st):
                        o = feature_work(pcs, prop, store, precision, cents[0], gsets[0], G, recs)
                    kinds[G] = ((g_main, g_side, flat_a, flat_b), o)
                    continue
                pair, opair = [], []
                for w in (0, 1):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=st):
                        o = pipelined_step(pcs, pcs, prop, store, precision, cents[w], cents[1 - w], side, G)
                    pair.append(g); opair.append(o)
                kinds[G] = (pair, opair)
            graphs.append(kinds); outs.append(None)
    elif not args.no_graph:
        for st, geo in zip(streams, geos):
            with torch.cuda.stream(st):
                step(pcs, prop, store, precision, geo)             # lazy init + allocator warm-up
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                o = step(pcs, prop, store, precision, geo)
            graphs.append(g); outs.append(o)

    pending = []          # own records of the replays since the last all-gather
    replays = [{group: 0, 1: 0} for _ in streams]      # ping-pong position per unit size (single-graph mode)

    def run_steps(k, timer=None):
        prev = ops.set_timer(timer)
        try:
            if graphs and timer is None and pipelined:
                units = [group] * (k // group) + [1] * (k % group)
                for u, G in enumerate(units):
                    j = u % len(streams)
                    if two_graph:
                        (g_main, g_side, flat_a, flat_b), o = graphs[j][G]
                        # side graph r overwrites set B, which the closing copy of replay r-1 read;
                        # main graph r reads set A, which that copy wrote
                        sides[j].wait_stream(streams[j])
                        with torch.cuda.stream(sides[j]):
                            g_side.replay()
                    else:
                        w = replays[j][G] & 1
                        replays[j][G] += 1
                        pair, opair = graphs[j][G]
                        o = opair[w]
                    with torch.cuda.stream(streams[j]):
                        if two_graph:
                            g_main.replay()
                            streams[j].wait_stream(sides[j])
                            flat_a.copy_(flat_b)
                        else:
                            pair[w].replay()
                        # The records of the replay are copied aside (the graph rewrites its own); ONE
                        # all-gather moves all of them at the fence that closes the timed region.  A
                        # collective per replay cost 16 % at one rank (measured, waiting for it or not:
                        # RCCL's stream is a fifth stream on HIP's four hardware queues, and the event it
                        # waits on parks in a queue that a step stream shares; more queues are slower
                        # still: 7.0-7.6k scenes/s at 5 / 6 / 8), and xGMI rewards fewer, larger
                        # collectives anyway.
                        recs = [rec for _, rec in o]
                        if dist.is_initialized():
                            pending.append(torch.cat(recs, 0) if len(recs) > 1 else recs[0].clone())
                            if len(pending) >= 256:
                                flush_records()
                return
            for i in range(k):
                j = i % len(streams)
                with torch.cuda.stream(streams[j]):
                    if graphs and timer is None:
                        graphs[j].replay()
                        rec = outs[j][1]
                    else:
                        rec = step(pcs, prop, store, precision, None if timer is not None else geos[j])[1]
                    pdist.all_gather_detections(rec)
        finally:
            ops.set_timer(prev)

    def flush_records():
        """One all-gather of every record produced since the last flush (ordered by scene id)."""
        if pending:
            torch.cuda.synchronize()                    # the copies were made on the step streams
            gathered = pdist.all_gather_detections(torch.cat(pending, 0))
            pending.clear()
            return gathered

    def fence():
        flush_records()
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps)
    t_issue = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # Instrumented pass (eager, HIP events around every launch, same streams and steps):
    # per-kernel durations for the roofline numbers.  Not part of `value`.
    timer = ops.KernelTimer()
    one = streams[:1]
    streams, one = one, streams          # instrumented pass: ONE stream, so no kernel overlaps another
    run_steps(min(args.steps, 4), timer)
    fence()
    streams = one

    if rank == 0:
        scenes = args.batch * args.steps * world
        ksum = timer.summary()
        kernels = {}
        for name, d in ksum.items():
            e = {"launches": d["launches"], "ms_total": round(d["ms"], 3),
                 "avg_us": round(1e3 * d["ms"] / d["launches"], 2)}
            if "flops" in d:
                peak = roofline.FP32_MFMA_PEAK_TFLOPS if precision == 0 else roofline.BF16_MFMA_PEAK_TFLOPS
                e.update(bound="mfma", achieved=d["flops"] / (d["ms"] * 1e-3) / 1e12, peak=peak, unit="TFLOP/s")
            elif name in ("group_point", "three_interpolate", "three_nn_interpolate", "gather_point", "sa_group_add_layer"):
                e.update(bound="hbm", achieved=d["bytes"] / (d["ms"] * 1e-3) / 1e9, peak=roofline.HBM_PEAK_GBS,
                         unit="GB/s")
            if "achieved" in e:
                e["frac"] = e["achieved"] / e["peak"]
                e["achieved"] = round(e["achieved"], 3); e["frac"] = round(e["frac"], 4)
            if "tests" in d:
                e["gtests_per_s"] = round(d["tests"] / (d["ms"] * 1e-3) / 1e9, 2)
            kernels[name] = e
        # dominant roofline-bound kernel = the single (kernel, shape) with the most time in the
        # step among the HBM- / MFMA-bound ones
        best = None
        for (name, _), d in timer.summary(by_shape=True).items():
            if "flops" in d:
                peak = roofline.FP32_MFMA_PEAK_TFLOPS if precision == 0 else roofline.BF16_MFMA_PEAK_TFLOPS
                cand = dict(kernel=name, shape=d.get("shape"), bound="mfma", unit="TFLOP/s", peak=peak,
                            achieved=d["flops"] / (d["ms"] * 1e-3) / 1e12, algorithmic=d["flops"] // d["launches"])
            elif name in ("group_point", "three_interpolate", "sa_group_add_layer"):
                cand = dict(kernel=name, shape=d.get("shape"), bound="hbm", unit="GB/s", peak=roofline.HBM_PEAK_GBS,
                            achieved=d["bytes"] / (d["ms"] * 1e-3) / 1e9, algorithmic=d["bytes"] // d["launches"])
            else:
                continue
            cand.update(ms=d["ms"], launches=d["launches"], avg_launch_us=round(1e3 * d["ms"] / d["launches"], 2))
            if best is None or cand["ms"] > best["ms"]:
                best = cand
        roof = None
        if best:
            traffic, tsrc = None, None
            # HBM bytes per launch from the committed PMC passes (profiles/, same kernel and shape)
            pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_hot_kernels.json")
            pmc_keys = {   # (kernel family, shape) -> entry of the PMC summary (tools/pmc_probe.py launches these shapes)
                ("mlp_layer", "rows65536 K608 N256 pool1"): "mlp_kernel<2, 2, 2, 2, false, false, 16> grid_threads=262144",
                ("mlp_layer", "rows262144 K260 N128 pool1"): "mlp_kernel<2, 2, 2, 2, false, false, 16> grid_threads=524288",
                ("mlp_layer", "rows524288 K96 N128 pool32"): "mlp_kernel<2, 2, 2, 2, false, false, 16> grid_threads=1048576",
                ("sa_chain", "rows2097152 K4 N32-32-64 pool32"): "sa_chain_kernel<4, 32, 32, 64, 32, 32>",
                ("sa_chain2", "rows524288 N64-96-128 pool32"): "sa_chain2_kernel<96, 6>",
                ("mlp_layer", "rows4096 K512 N512 pool1"): "mlp_kernel<4, 1, 1, 2, false, false, 32, false> grid_threads=65536",
            }
            pk = pmc_keys.get((best["kernel"], best["shape"]))
            if pk and precision == 0 and args.batch == BATCH and os.path.exists(pmc_path):
                pm = json.load(open(pmc_path)).get(pk)
                if pm and "hbm_bytes_corrected" in pm:
                    traffic = pm["hbm_bytes_corrected"]
                    tsrc = ("profiles/r01_pmc_hot_kernels.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                            "tools/pmc_probe.py on this kernel and shape, read side x2 (gfx950); raw %d B"
                            % int(pm["hbm_bytes_raw"]))
            roof = {"kernel": best["kernel"], "shape": best["shape"], "bound": best["bound"],
                    "achieved": round(best["achieved"], 3), "peak": best["peak"], "unit": best["unit"],
                    "frac": round(best["achieved"] / best["peak"], 4), "traffic": traffic,
                    "traffic_source": tsrc, "algorithmic_per_launch": best["algorithmic"],
                    "avg_launch_us": best["avg_launch_us"], "launches": best["launches"]}
        fps = None
        if "fps" in ksum:
            d = ksum["fps"]
            # per launch all clouds of the batch run concurrently (one workgroup each)
            fps = {"bound": "latency", "us_per_iteration": round(1e3 * d["ms"] / (d["iters"]), 4),
                   "ms_total": round(d["ms"], 3), "launches": d["launches"],
                   "gdist_evals_per_s": round(d["dist_evals"] / (d["ms"] * 1e-3) / 1e9, 2)}
        nms_ms, nms_keep = time_nms(device)
        hbm_copy = time_hbm_copy(device)
        out = {
            "metric": "SA/FP-stack scenes/sec (16k pts)", "value": round(scenes / elapsed, 2), "unit": "scenes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "PointRCNN RPN SA+FP stack, batch %d x %d pts per GPU, fused group->shared-MLP"
                                   "->max-pool (MFMA), + rotated NMS %d->100 per scene%s" %
                                   (args.batch, NPTS, NPROP, " + RCCL all-gather of detections" if world > 1 else ""),
                       "batch_per_gpu": args.batch, "points": NPTS, "streams": len(streams),
                       "hip_graph": bool(graphs), "host_issue_ms_per_step": round(1e3 * t_issue / args.steps, 3),
                       "geometry_streams": bool(args.geo_stream), "fps_one_step_ahead": bool(pipelined),
                       "duplicate_free_groups": bool(PU.SA_DEDUP), "ball_fill": ball_fill(pcs),
                       "geometry_one_replay_ahead": bool(pipelined and args.geo_ahead), "batches_per_replay": group,
                       "kernel_timing": "HIP events around every launch in an eager, single-stream instrumented "
                                        "pass of the same steps after the timed region",
                       "parallelism": "scene-sharded x%d" % world,
                       "collective": "one RCCL all-gather of all detection records per timed region" if dist.is_initialized() else None},
            "roofline": roof, "kernels": kernels, "fps": fps,
            "nms": {"ms_9000_to_100": round(nms_ms, 4), "kept": nms_keep, "pairs": 9000 * 8999 // 2,
                    "metric": "rotated-NMS ms @9k boxes"},
            "fps_dense": time_fps_dense(device),
            "frustum": time_frustum(device, store, precision),
            "hbm_copy_gbps": {"value": round(hbm_copy, 1), "what": "torch device-to-device copy of 1 GiB, read+write bytes / time"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(store)
        print(json.dumps(out))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
