#!/usr/bin/env python3
"""Largest FP level (16384 <- 4096, C=128, batch 16): the fused culled search + interpolation against its
two halves launched separately (run under rocprofv3 --kernel-trace --stats for per-kernel durations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointrcnn_amd import ops, synthetic as S

dev = torch.device("cuda:0")
b = 16
pc = torch.as_tensor(S.kitti_batch(0, b)).to(dev)
xyz = pc[:, :, :3].contiguous()
new1 = ops.gather_point(xyz, ops.farthest_point_sample(4096, xyz))
g = torch.Generator(device="cpu"); g.manual_seed(0)
f128 = torch.randn(b, 4096, 128, generator=g).to(dev)
u4 = torch.randn(4, 128, generator=g).mul_(0.1).to(dev)
t128 = torch.zeros(128, device=dev)
for _ in range(5):
    ops.three_nn_interpolate_affine(xyz, new1, f128, pc, u4, t128)
    d, i = ops.three_nn(xyz, new1)
    w = ops.fp_weights(d)
    ops.three_interpolate_affine(f128, i, w, pc, u4, t128)
torch.cuda.synchronize()
