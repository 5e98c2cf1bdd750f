#!/usr/bin/env python3
"""One dense layer launched on two streams at once, many times: are the outputs bit-identical to a launch alone?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointrcnn_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu"); g.manual_seed(0)
for prec, name in ((ops.PRECISION_BF16, "bf16"), (ops.PRECISION_FP32, "f32")):
    for rows, K, N in ((262144, 128, 128), (65536, 256, 128), (65536, 256, 256)):
        x = torch.randn(rows, K, generator=g).to(dev); w = torch.randn(K, N, generator=g).mul_(0.1).to(dev)
        s = torch.ones(N, device=dev); t = torch.zeros(N, device=dev)
        ref = ops.mlp_layer(x, w, s, t, precision=prec).clone()
        torch.cuda.synchronize()
        st = [torch.cuda.Stream(device=dev) for _ in range(2)]
        bad = 0
        for r in range(30):
            outs = []
            for q in st:
                with torch.cuda.stream(q):
                    outs.append(ops.mlp_layer(x, w, s, t, precision=prec))
                    outs.append(ops.mlp_layer(x, w, s, t, precision=prec))
            torch.cuda.synchronize()
            bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
        print(name, rows, K, N, "differing outputs: %d of 120" % bad, flush=True)
