#!/usr/bin/env python3
"""Per-launch table of ONE eager, single-stream bench step (batch 16 x 16384): every
(kernel family, shape) with its HIP-event duration and algorithmic rate, sorted by time.
  python tools/step_table.py [--precision f32|bf16] [--reps 5]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

import bench
from pointrcnn_amd import ops, pointnet_util as PU


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    prec = ops.PRECISION_FP32 if args.precision == "f32" else ops.PRECISION_BF16
    store = PU.ParamStore(seed=7, device=dev)
    pcs, prop = bench.make_inputs(0, dev, bench.BATCH)
    for _ in range(2):
        bench.step(pcs, prop, store, prec)
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    for _ in range(args.reps):
        bench.step(pcs, prop, store, prec)
    ops.set_timer(None)
    rows = []
    for (name, _), d in timer.summary(by_shape=True).items():
        us = 1e3 * d["ms"] / args.reps
        rate = ""
        if "flops" in d:
            rate = "%6.1f TF/s" % (d["flops"] / args.reps / (us * 1e-6) / 1e12)
        elif "bytes" in d:
            rate = "%6.0f GB/s" % (d["bytes"] / args.reps / (us * 1e-6) / 1e9)
        rows.append((us, name, d.get("shape", ""), d["launches"] // args.reps, rate))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print("# one step, batch %d, %s: %.1f us of kernels (eager, single stream)" % (bench.BATCH, args.precision, tot))
    for us, name, shape, n, rate in rows:
        print("%8.1f us  x%-2d %-22s %-40s %s" % (us, n, name, shape, rate))


if __name__ == "__main__":
    main()
