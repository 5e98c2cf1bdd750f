"""Generates tests/golden/*.npz (TEST INFRASTRUCTURE ONLY; run in the build container).

Vectors whose `source` field says "reference" are outputs of the reference's OWN
standalone CPU sources compiled into oracle/_ref by oracle/Makefile (query_ball_point.cpp,
interpolate.cpp) on seeded inputs.  The reference tree does not exist on the GPU box, so
these small fixtures are what pins the oracle (and through it the HIP kernels) there.
Vectors with source "transcription" come from oracle/fps_numpy.py (the reference has no
CPU farthest-point-sampling; parity unpinned at the kernel level).
Run:  python -m oracle.make_golden
"""
import os

import numpy as np

from . import fps_numpy, ref

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    if not ref.available():
        raise SystemExit("oracle/_ref is not built (make -C oracle ref)")
    rng = np.random.default_rng(20251110)

    # --- query_ball_point_cpu + group_point_cpu (query_ball_point.cpp:19-66) --------------
    b, n, m, ns, c = 2, 600, 70, 16, 7
    xyz1 = rng.random((b, n, 3), dtype=np.float32)
    xyz1[:, 100:140] = xyz1[:, 0:40]                      # exact duplicates
    xyz2 = xyz1[:, rng.choice(n, m, replace=False)].copy()  # centroids are points (d clamps to 1e-20)
    pts = rng.random((b, n, c), dtype=np.float32)
    cases = {}
    for r in (0.05, 0.15, 0.4):
        idx = ref.query_ball_point_cpu(r, ns, xyz1, xyz2)
        cases["idx_r%g" % r] = idx
    grouped = ref.group_point_cpu(pts, cases["idx_r0.15"])
    np.savez_compressed(os.path.join(OUT, "grouping_ref.npz"), source="reference",
                        xyz1=xyz1, xyz2=xyz2, points=pts, nsample=ns, radii=np.array([0.05, 0.15, 0.4]),
                        grouped_r0p15=grouped, **{k.replace(".", "p"): v for k, v in cases.items()})

    # radius exactly equal to a pair distance: strict '<' must exclude it
    a = np.zeros((1, 4, 3), np.float32)
    a[0, 1] = [0.5, 0, 0]; a[0, 2] = [0.25, 0, 0]; a[0, 3] = [3, 4, 0]
    q = np.zeros((1, 1, 3), np.float32)
    np.savez_compressed(os.path.join(OUT, "ballquery_edge_ref.npz"), source="reference", xyz1=a, xyz2=q,
                        idx_r0p5=ref.query_ball_point_cpu(0.5, 4, a, q),
                        idx_r5=ref.query_ball_point_cpu(5.0, 4, a, q),
                        idx_r5p0001=ref.query_ball_point_cpu(np.nextafter(np.float32(5), np.float32(6)), 4, a, q))

    # --- interpolate_cpu (interpolate.cpp:84-104 == tf_interpolate.cpp:107-127) -----------
    b, n, m, c = 2, 300, 40, 20
    points = rng.standard_normal((b, m, c)).astype(np.float32)
    idx = rng.integers(0, m, (b, n, 3)).astype(np.int32)
    w = rng.random((b, n, 3), dtype=np.float32)
    w /= w.sum(-1, keepdims=True)
    out = ref.interpolate_cpu(points, idx, w)
    np.savez_compressed(os.path.join(OUT, "interpolate_ref.npz"), source="reference", points=points,
                        idx=idx, weight=w, out=out)

    # --- threenn_cpu origin variant (interpolate.cpp:21-64) --------------------------------
    xyz2 = rng.standard_normal((3, 50, 3)).astype(np.float32)
    xyz2[1, 7] = xyz2[1, 3]                               # exact tie: earlier index first
    xyz2[2, 2:] = 0                                       # m effectively tiny: many ties at 0
    d, i = ref.threenn_origin_cpu(5, xyz2)
    d1, i1 = ref.threenn_origin_cpu(2, xyz2[:, :1])        # m = 1
    d2, i2 = ref.threenn_origin_cpu(2, xyz2[:, :2])        # m = 2
    np.savez_compressed(os.path.join(OUT, "threenn_origin_ref.npz"), source="reference", xyz2=xyz2,
                        dist=d, idx=i, dist_m1=d1, idx_m1=i1, dist_m2=d2, idx_m2=i2)

    # --- FPS: NumPy transcription of tf_sampling_g.cu:105-170 ------------------------------
    base = rng.standard_normal((300, 3)).astype(np.float32)
    cloud = base[rng.integers(0, 300, (2, 1500))]          # duplicates; n not a multiple of 512
    idx_dup = fps_numpy.farthest_point_sample(400, cloud)   # m > #distinct points
    tie = np.zeros((1, 1024, 3), np.float32)
    tie[0, 2] = [1, 0, 0]; tie[0, 513] = [1, 0, 0]          # tied maxima at k=2 and k=513
    idx_tie = fps_numpy.farthest_point_sample(4, tie)
    small = rng.standard_normal((3, 100, 3)).astype(np.float32)
    idx_small = fps_numpy.farthest_point_sample(100, small)
    np.savez_compressed(os.path.join(OUT, "fps_transcription.npz"), source="transcription",
                        cloud=cloud, idx_dup=idx_dup, tie=tie, idx_tie=idx_tie, small=small,
                        idx_small=idx_small)
    grad_golden()
    knn_golden()
    print("golden vectors written to", os.path.normpath(OUT))


def knn_golden():
    """selection_sort_cpu (selection_sort.cpp:20-63) on small matrices with many exact ties
    (the tie order is the part that needs pinning) and on a tie-free one."""
    rng = np.random.default_rng(20251112)
    d_ties = rng.integers(0, 6, (2, 5, 40)).astype(np.float32)          # heavy ties
    d_ties[0, 0, :] = 3.0                                               # a constant row
    d_ties[1, 1, :8] = 9.0                                              # large values in front: displaced
    d_free = rng.random((1, 4, 64), dtype=np.float32)
    out = {}
    for name, d, k in (("ties_k3", d_ties, 3), ("ties_k16", d_ties, 16), ("ties_k40", d_ties, 40),
                       ("free_k10", d_free, 10)):
        idx, val = ref.selection_sort_cpu(k, d)
        out[name + "_idx"] = idx
        out[name + "_val"] = val
    np.savez_compressed(os.path.join(OUT, "selection_sort_ref.npz"), source="reference", d_ties=d_ties,
                        d_free=d_free, **out)


def grad_golden():
    """Backward ops: group_point_grad_cpu (query_ball_point.cpp:70-84) and
    interpolate_grad_cpu (interpolate.cpp:108-130) on seeded inputs (own RNG stream, so the
    forward fixtures above never change)."""
    rng = np.random.default_rng(20251111)
    b, n, m, ns, c = 2, 500, 60, 16, 9
    idx = rng.integers(0, n, (b, m, ns)).astype(np.int32)
    idx[0, :, 3:] = idx[0, :, :1]                          # heavy rows: one point collects many entries
    idx[1, 5] = 7
    grad_out = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    gp = ref.group_point_grad_cpu(n, idx, grad_out)
    b2, n2, m2, c2 = 2, 400, 50, 12
    idx3 = rng.integers(0, m2, (b2, n2, 3)).astype(np.int32)
    idx3[1, :, 1] = idx3[1, :, 0]                          # the same target twice within one point
    w = rng.random((b2, n2, 3), dtype=np.float32)
    w /= w.sum(-1, keepdims=True)
    g3 = rng.standard_normal((b2, n2, c2)).astype(np.float32)
    gi = ref.interpolate_grad_cpu(m2, idx3, w, g3)
    np.savez_compressed(os.path.join(OUT, "grad_ref.npz"), source="reference", n=n, idx=idx,
                        grad_out=grad_out, grad_points=gp, m3=m2, idx3=idx3, weight3=w, grad_out3=g3,
                        grad_points3=gi)


if __name__ == "__main__":
    import sys
    if "--grad-only" in sys.argv:
        grad_golden()
        raise SystemExit(0)
    if "--knn-only" in sys.argv:
        knn_golden()
        raise SystemExit(0)
    main()
