#!/usr/bin/env python3
"""Is the stack deterministic run to run, alone and with a second stream busy?  Reports the first level
whose tensor differs from the first run's.  python tools/probe_determinism.py [f32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from pointrcnn_amd import ops, pointnet_util as PU, stacks

prec = ops.PRECISION_BF16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else ops.PRECISION_FP32
dev = torch.device("cuda:0")
store = PU.ParamStore(seed=7, device=dev)
pcs, prop = bench.make_inputs(0, dev, bench.BATCH)

def run():
    f, aux = stacks.rpn_sa_fp_stack(pcs, params=store, precision=prec)
    return [("xyz%d" % i, t) for i, t in enumerate(aux["xyz"][1:], 1)] + \
           [("sa%d" % i, t) for i, t in enumerate(aux["sa_feats"][1:], 1)] + \
           [("fp%d" % i, t) for i, t in enumerate(aux["fp_feats"])]

def diff(ref, got, tag):
    bad = [(n, int((a != b).sum().item()), float((a - b).abs().max().item())) for (n, a), (_, b) in zip(ref, got) if not torch.equal(a, b)]
    print(tag, "identical" if not bad else "DIFFERS: %s" % bad[:4], flush=True)
    return not bad

ref = [(n, t.clone()) for n, t in run()]
torch.cuda.synchronize()
for i in range(4):
    got = run(); torch.cuda.synchronize()
    diff(ref, got, "alone run %d:" % i)
s2 = torch.cuda.Stream(device=dev)
for i in range(6):
    with torch.cuda.stream(s2):
        other = run()
    got = run()
    torch.cuda.synchronize()
    diff(ref, got, "with a second stream busy, run %d (main):" % i)
    diff(ref, other, "with a second stream busy, run %d (other):" % i)
