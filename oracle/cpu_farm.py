"""N-core leg of bench.py's cpu_baseline (TEST / MEASUREMENT INFRASTRUCTURE ONLY, like the rest of
oracle/): scenes farmed over worker processes, one BLAS thread each, every worker running the
single-threaded oracle stack (the reference's CPU ops are single-threaded: compile.sh:1,
tf_interpolate_compile.sh:5) on its own scenes.  SURVEY.md 8(d): "an N-core number (scenes farmed
over processes) with N stated".  Workers are spawned (never forked: the parent holds a GPU context)
and import NumPy only."""
import os
import time


def _worker(args):
    first_scene, n_scenes, npts, params = args
    os.environ["OMP_NUM_THREADS"] = "1"
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                           # pragma: no cover
        threadpool_limits = None
    import numpy as np  # noqa: F401
    from oracle import oracle as O
    from oracle import stack as ostack
    from pointrcnn_amd import synthetic as S
    O.lib()
    clouds = [S.kitti_batch(first_scene + i, 1, npts) for i in range(n_scenes)]
    boxes, scores = S.nms_boxes(seed=7000 + first_scene, N=2048, clusters=40)
    t0 = time.perf_counter()
    for pc in clouds:
        if threadpool_limits:
            with threadpool_limits(limits=1):
                ostack.rpn_sa_fp_stack(pc, params, gemm='blas')
        else:
            ostack.rpn_sa_fp_stack(pc, params, gemm='blas')
        O.nms_rotate_cpu(boxes, scores, 0.8, 100)
    return t0, time.perf_counter()


def farm(params, cores, scenes_per_worker=2, npts=16384):
    """Returns (scenes/s over all workers, scenes run): wall time from the first worker's start of
    compute to the last worker's end (process start-up and input generation excluded)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    jobs = [(1000 * (w + 1), scenes_per_worker, npts, params) for w in range(cores)]
    with ctx.Pool(cores) as pool:
        spans = pool.map(_worker, jobs, chunksize=1)
    t0 = min(s[0] for s in spans)
    t1 = max(s[1] for s in spans)
    n = cores * scenes_per_worker
    return n / (t1 - t0), n
