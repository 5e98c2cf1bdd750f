#!/usr/bin/env python3
"""Level-1 and level-2 ball queries of the RPN stack (batch 16) for a kernel trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointrcnn_amd import ops, synthetic as S
dev = torch.device("cuda:0")
b = 16
pc = torch.as_tensor(S.kitti_batch(0, b)).to(dev)
xyz = pc[:, :, :3].contiguous()
new1 = ops.gather_point(xyz, ops.farthest_point_sample(4096, xyz))
new2 = ops.gather_point(new1, ops.farthest_point_sample(1024, new1))
for _ in range(5):
    ops.query_ball_point_msg([0.1, 0.5], [16, 32], xyz, new1)
    ops.query_ball_point_msg([0.5, 1.0], [16, 32], new1, new2)
torch.cuda.synchronize()
