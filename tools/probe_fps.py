import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointrcnn_amd import ops, synthetic as S
dev = torch.device("cuda:0")
pc = torch.as_tensor(S.kitti_batch(0, 16)).to(dev)[:, :, :3].contiguous()
x = pc
for m in (4096, 1024, 256, 64):
    idx = ops.farthest_point_sample(m, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): idx = ops.farthest_point_sample(m, x)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 3
    print("H1=%s n %d -> %d: %.1f us  %.3f us/iter  checksum %d" % (os.environ.get("PRCNN_FPS_H1"), x.shape[1], m, us, us / (m - 1), int(idx.long().sum())), flush=True)
    x = ops.gather_point(x, idx)
