#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the RPN stack's shapes (one stream, HIP events).

  python tools/kbench.py [--batch 16] [--iters 20] [--only fps,ball,...]
Prints one line per (kernel, shape): average launch us and the achieved fraction of the
roofline that bounds it (SURVEY.md 8d formulas).  Used to iterate on single kernels and
as the command behind the per-kernel rocprof summaries in profiles/.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from pointrcnn_amd import ops, synthetic as S
from pointrcnn_amd.nms_rotate import nms_rotate_batched
from pointrcnn_amd.stacks import RPN_FP, RPN_SA


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters      # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--precision", type=int, default=0)
    a = ap.parse_args()
    only = set(a.only.split(",")) if a.only else None
    want = lambda k: only is None or k in only
    dev = torch.device("cuda:0")
    b = a.batch
    pc = torch.as_tensor(S.kitti_batch(0, b)).to(dev)
    xyz = pc[:, :, :3].contiguous()
    feats = pc[:, :, 3:].contiguous()
    rng = torch.Generator(device="cpu"); rng.manual_seed(0)
    rows = []

    def rep(name, shape, us, frac=None, extra=""):
        rows.append((name, shape, us, frac, extra))
        print("%-22s %-34s %10.1f us  %s %s" % (name, shape, us, "" if frac is None else "frac=%.3f" % frac, extra),
              flush=True)

    levels = [(xyz, feats)]
    for li, (m, radii, nss, mlps) in enumerate(RPN_SA):
        x, f = levels[-1]
        n = x.shape[1]
        c = f.shape[2]
        if want("fps"):
            us = timeit(lambda: ops.farthest_point_sample(m, x), max(3, a.iters // 4))
            rep("fps", "b%d n%d m%d" % (b, n, m), us, None, "%.3f us/iter" % (us / (m - 1)))
        idx_fps = ops.farthest_point_sample(m, x)
        new_xyz = ops.gather_point(x, idx_fps)
        outs = []
        for r, ns, mlp in zip(radii, nss, mlps):
            if want("ball"):
                us = timeit(lambda: ops.query_ball_point(r, ns, x, new_xyz), a.iters)
                rep("query_ball_point", "b%d n%d m%d r%g ns%d" % (b, n, m, r, ns), us, None,
                    "%.1f Gtests/s" % (b * m * n / us / 1e3))
            idx, _ = ops.query_ball_point(r, ns, x, new_xyz)
            if want("group"):
                us = timeit(lambda: ops.group_point(f, idx), a.iters)
                by = b * m * ns * (4 + 8 * c)
                rep("group_point", "b%d m%d ns%d C%d" % (b, m, ns, c), us, by / us / 1e3 / 8000, "%.0f GB/s" % (by / us / 1e3))
            if want("knn") and r == radii[-1]:
                us = timeit(lambda: ops.knn_point(ns, x, new_xyz), max(3, a.iters // 4))
                rep("knn_point", "b%d n%d m%d k%d" % (b, n, m, ns), us, None, "%.1f Gdist/s (5 passes)" % (5 * b * m * n / us / 1e3))
            if want("grad"):
                go = torch.randn(b, m, ns, c, generator=rng).to(dev)
                us = timeit(lambda: ops.group_point_grad(f, idx, go), a.iters)
                by = b * m * ns * (4 + 4 * c) + b * n * c * 4
                rep("group_point_grad", "b%d n%d m%d ns%d C%d" % (b, n, m, ns, c), us, by / us / 1e3 / 8000, "%.0f GB/s" % (by / us / 1e3))
            cin = c + 3
            W = [(torch.randn(ci, co, generator=rng).mul_(0.1).to(dev), torch.ones(co, device=dev), torch.zeros(co, device=dev))
                 for ci, co in zip([cin] + mlp[:-1], mlp)]
            if want("mlp") and ops.sa_chain_supported(c, mlp, ns, a.precision):
                us = timeit(lambda: ops.sa_group_mlp3_pool(x, f, new_xyz, idx, W), a.iters)
                fl = 2 * b * m * ns * (cin * mlp[0] + mlp[0] * mlp[1] + mlp[1] * mlp[2])
                rep("sa_chain (3 layers)", "rows%d K%d N%s" % (b * m * ns, cin, mlp), us, fl / us / 1e6 / 157.3, "%.1f TF" % (fl / us / 1e6))
            if want("mlp"):
                us = timeit(lambda: ops.sa_group_mlp_layer(x, f, new_xyz, idx, *W[0], precision=a.precision), a.iters)
                fl = 2 * b * m * ns * cin * mlp[0]
                rep("sa_group_mlp_layer", "rows%d K%d N%d" % (b * m * ns, cin, mlp[0]), us, fl / us / 1e6 / 157.3, "%.1f TF" % (fl / us / 1e6))
            h = ops.sa_group_mlp_layer(x, f, new_xyz, idx, *W[0], precision=a.precision).reshape(-1, mlp[0])
            for j in (1, 2):
                pool = ns if j == 2 else 1
                if want("mlp"):
                    us = timeit(lambda: ops.mlp_layer(h, *W[j], pool=pool, precision=a.precision), a.iters)
                    fl = 2 * h.shape[0] * W[j][0].shape[0] * W[j][0].shape[1]
                    rep("mlp_layer%s" % ("+pool" if pool > 1 else ""), "rows%d K%d N%d" % (h.shape[0], W[j][0].shape[0], W[j][0].shape[1]),
                        us, fl / us / 1e6 / 157.3, "%.1f TF" % (fl / us / 1e6))
                h = ops.mlp_layer(h, *W[j], pool=pool, precision=a.precision)
            outs.append(h.reshape(b, m, -1))
        levels.append((new_xyz, torch.cat(outs, -1)))
    # FP levels
    c2 = levels[4][1]
    for k, mlp in zip((3, 2, 1, 0), RPN_FP):
        x1 = levels[k][0]; x2 = levels[k + 1][0]
        p2 = c2 if k == 3 else torch.randn(b, x2.shape[1], prev_c, generator=rng).to(dev)
        n1, m2, c = x1.shape[1], x2.shape[1], p2.shape[2]
        if want("three_nn"):
            us = timeit(lambda: ops.three_nn(x1, x2), a.iters)
            rep("three_nn", "b%d n%d m%d" % (b, n1, m2), us, None, "%.1f Gtests/s" % (b * n1 * m2 / us / 1e3))
        d, i3 = ops.three_nn(x1, x2)
        w = ops.fp_weights(d)
        if want("interp"):
            us = timeit(lambda: ops.three_interpolate(p2, i3, w), a.iters)
            by = b * n1 * (24 + 16 * c)
            rep("three_interpolate", "b%d n%d C%d" % (b, n1, c), us, by / us / 1e3 / 8000, "%.0f GB/s" % (by / us / 1e3))
            us = timeit(lambda: ops.three_nn_interpolate(x1, x2, p2), a.iters)
            by = b * (12 * n1 + 12 * m2) + b * n1 * 16 * c
            rep("three_nn_interpolate", "b%d n%d m%d C%d" % (b, n1, m2, c), us, by / us / 1e3 / 8000, "%.0f GB/s" % (by / us / 1e3))
        if want("grad"):
            go = torch.randn(b, n1, c, generator=rng).to(dev)
            us = timeit(lambda: ops.three_interpolate_grad(p2, i3, w, go), a.iters)
            by = b * n1 * (24 + 12 * c) + b * m2 * c * 4
            rep("three_interpolate_grad", "b%d n%d m%d C%d" % (b, n1, m2, c), us, by / us / 1e3 / 8000, "%.0f GB/s" % (by / us / 1e3))
        c1 = levels[k][1].shape[2] + (3 if k == 0 else 0)
        cin = c + c1
        if want("mlp"):
            xin = torch.randn(b * n1, cin, generator=rng).to(dev)
            for co in mlp:
                Wt = torch.randn(cin, co, generator=rng).mul_(0.05).to(dev)
                us = timeit(lambda: ops.mlp_layer(xin, Wt, torch.ones(co, device=dev), torch.zeros(co, device=dev), precision=a.precision), a.iters)
                fl = 2 * b * n1 * cin * co
                rep("mlp_layer(fp)", "rows%d K%d N%d" % (b * n1, cin, co), us, fl / us / 1e6 / 157.3, "%.1f TF" % (fl / us / 1e6))
                xin = torch.randn(b * n1, co, generator=rng).to(dev); cin = co
        prev_c = mlp[-1]
    if want("nms"):
        for N, B in ((9000, 1), (2048, 1), (2048, 16)):
            bx, sc = S.nms_boxes(seed=9000, N=N, clusters=max(2, N // 60))
            bt = torch.as_tensor(np.stack([bx] * B)).to(dev); st = torch.as_tensor(np.stack([sc] * B)).to(dev)
            us = timeit(lambda: nms_rotate_batched(bt, st, 0.8, 100), a.iters)
            rep("nms_rotate", "B%d N%d ->100" % (B, N), us, None, "%.3f ms" % (us / 1e3))
    tot = {}
    for name, shape, us, frac, _ in rows:
        tot[name] = tot.get(name, 0) + us
    print("---- sum of average launch times per kernel family (us), batch %d ----" % b)
    for k, v in sorted(tot.items(), key=lambda t: -t[1]):
        print("%-24s %10.1f" % (k, v))


if __name__ == "__main__":
    main()
