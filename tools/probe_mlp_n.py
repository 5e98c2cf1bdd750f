import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pointrcnn_amd import ops
dev = torch.device("cuda:0")
class CR: pass
def run(rows_cap, rows, cin, cout, prec):
    cr = CR(); cr.capacity = rows_cap; cr.rows = torch.tensor([rows], dtype=torch.int32, device=dev)
    x = torch.randn(rows_cap, cin, device=dev); w = torch.randn(cin, cout, device=dev) * 0.1
    s = torch.ones(cout, device=dev); t = torch.zeros(cout, device=dev)
    for _ in range(2): ops.mlp_layer_n(cr, x, w, s, t, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.mlp_layer_n(cr, x, w, s, t, precision=prec)
    e1.record(); torch.cuda.synchronize()
    print("cap %d rows %d K%d N%d prec %d: %.1f us" % (rows_cap, rows, cin, cout, prec, 1e3 * e0.elapsed_time(e1) / 5), flush=True)
for prec in (0, 1):
    for cout in (64, 96, 128):
        run(524288, 60000, 64, cout, prec)
    run(524288, 524288, 64, 96, prec)
    run(32768, 3000, 256, 384, prec)
