import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointrcnn_amd import ops
dev = torch.device("cuda:0")
def run(rows, cin, cout, prec=0):
    x = torch.randn(rows, cin, device=dev); w = torch.randn(cin, cout, device=dev) * 0.05
    s = torch.ones(cout, device=dev); t = torch.zeros(cout, device=dev)
    for _ in range(3): ops.mlp_layer(x, w, s, t, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.mlp_layer(x, w, s, t, precision=prec)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    print("%s rows %d K%d N%d: %.1f us  %.1f TF/s" % (os.environ.get("TAG", ""), rows, cin, cout, us, 2 * rows * cin * cout / us / 1e6), flush=True)
for shape in ((262144, 128, 128), (65536, 256, 256), (16384, 512, 512), (65536, 608, 256), (4096, 512, 512)):
    run(*shape)
