#!/usr/bin/env python3
"""Per-launch table of ONE eager pass of the three Frustum PointNet v2 stacks (BASELINE configs[1]:
cls + seg on 1024-point frustums, box net on 512 points, batch 32), like tools/step_table.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from pointrcnn_amd import ops, pointnet_util as PU, stacks, synthetic as S


def main():
    dev = torch.device("cuda", 0)
    batch, reps = 32, 5
    store = PU.ParamStore(seed=9, device=dev)
    fr = torch.as_tensor(S.frustum_batch(1, batch, 1024)).to(dev)
    obj = torch.as_tensor(S.frustum_batch(2, batch, 512, channels=3)).to(dev)
    onehot = torch.zeros((batch, 4), device=dev); onehot[:, 1] = 1

    def once():
        stacks.frustum_cls_stack(fr, params=store, precision=ops.PRECISION_FP32)
        stacks.frustum_seg_stack(fr, onehot, params=store, precision=ops.PRECISION_FP32)
        stacks.frustum_box_stack(obj, params=store, precision=ops.PRECISION_FP32)
    once(); once()
    torch.cuda.synchronize()
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    for _ in range(reps):
        once()
    ops.set_timer(None)
    rows = []
    for (name, _), d in timer.summary(by_shape=True).items():
        us = 1e3 * d["ms"] / reps
        rate = ""
        if "flops" in d:
            rate = "%6.1f TF/s" % (d["flops"] / reps / (us * 1e-6) / 1e12)
        elif "bytes" in d:
            rate = "%6.0f GB/s" % (d["bytes"] / reps / (us * 1e-6) / 1e9)
        rows.append((us, name, d.get("shape", ""), d["launches"] // reps, rate))
    rows.sort(reverse=True)
    print("# Frustum v2 cls + seg + box, batch %d, f32: %.1f us of kernels (eager, single stream)" % (batch, sum(r[0] for r in rows)))
    for us, name, shape, n, rate in rows:
        print("%8.1f us  x%-2d %-22s %-40s %s" % (us, n, name, shape, rate))


if __name__ == "__main__":
    main()
