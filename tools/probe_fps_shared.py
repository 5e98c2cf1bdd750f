"""Level-1 sampling (16 clouds x 16384 -> 4096): register-resident sampler against the small-footprint
one (prcnn_farthest_point_sample_shared), alone on the GPU.  PRCNN_FPS_SHARED_NW = 2 / 4 / 8 waves per cloud."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointrcnn_amd import ops, synthetic as S

dev = torch.device("cuda:0")
x = torch.as_tensor(S.kitti_batch(0, 16)).to(dev)[:, :, :3].contiguous()
ref = None
for shared in (False, True):
    idx = ops.farthest_point_sample(4096, x, shared=shared)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        idx = ops.farthest_point_sample(4096, x, shared=shared)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 3
    if ref is None:
        ref = idx
    print("shared=%s NW=%s: %.1f us  %.3f us/iter  equal %s" % (shared, os.environ.get("PRCNN_FPS_SHARED_NW"), us, us / 4095,
                                                                 bool(torch.equal(ref, idx))), flush=True)
