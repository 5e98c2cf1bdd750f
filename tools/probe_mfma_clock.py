#!/usr/bin/env python3
"""One dense and one few-row fp32 layer, launched REPS times each, for a GRBM_GUI_ACTIVE / SQ_INSTS_MFMA counter
pass and a kernel trace of the same script: elapsed clocks per launch / duration = the clock the kernel ran at,
SQ_INSTS_MFMA * 64 / (1024 SIMDs * elapsed clocks) = the share of matrix-core issue slots used."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointrcnn_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(0)
one = lambda c: (torch.ones(c, device=dev), torch.zeros(c, device=dev))
xa = torch.randn(262144, 128, generator=g).to(dev); wa = torch.randn(128, 128, generator=g).mul_(0.1).to(dev)
xb = torch.randn(4096, 512, generator=g).to(dev); wb = torch.randn(512, 512, generator=g).mul_(0.05).to(dev)
reps = int(os.environ.get("REPS", "20"))
for _ in range(reps):
    ops.mlp_layer(xa, wa, *one(128))
torch.cuda.synchronize()
for _ in range(reps):
    ops.mlp_layer(xb, wb, *one(512))
torch.cuda.synchronize()
