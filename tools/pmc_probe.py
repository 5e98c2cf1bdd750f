#!/usr/bin/env python3
"""One launch of each hot kernel at its largest RPN-stack shape (batch 16), for rocprofv3
counter passes (run each --pmc pass separately, see profiles/README.md):
  rocprofv3 --kernel-trace --stats ...            durations
  rocprofv3 --pmc FETCH_SIZE ...                  HBM read  (KB; x2 for wide coalesced reads on gfx950)
  rocprofv3 --pmc WRITE_SIZE ...                  HBM write (KB)
Each op is launched REPS times back to back; tools/pmc_summary.py averages per kernel name.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointrcnn_amd import ops, synthetic as S

REPS = 3


def main():
    dev = torch.device("cuda:0")
    b = 16
    g = torch.Generator(device="cpu"); g.manual_seed(0)
    pc = torch.as_tensor(S.kitti_batch(0, b)).to(dev)
    xyz = pc[:, :, :3].contiguous()
    feats = pc[:, :, 3:].contiguous()
    new1 = ops.gather_point(xyz, ops.farthest_point_sample(4096, xyz))
    new2 = ops.gather_point(new1, ops.farthest_point_sample(1024, new1))
    f1 = torch.randn(b, 4096, 96, generator=g).to(dev)
    f256 = torch.randn(b, 4096, 256, generator=g).to(dev)
    (idx2, _), = ops.query_ball_point_msg([1.0], [32], new1, new2)
    d, i3 = ops.three_nn(xyz, new1)
    w = ops.fp_weights(d)
    x260 = torch.randn(b * 16384, 260, generator=g).to(dev)
    W260 = torch.randn(260, 128, generator=g).mul_(0.05).to(dev)
    W99 = torch.randn(99, 64, generator=g).mul_(0.1).to(dev)
    x608 = torch.randn(b * 4096, 608, generator=g).to(dev)        # FP level 1, first layer: the step's most expensive launch
    W608 = torch.randn(608, 256, generator=g).mul_(0.05).to(dev)
    W4 = [(torch.randn(4, 32, generator=g).mul_(0.3).to(dev),) + (torch.ones(32, device=dev), torch.zeros(32, device=dev)),
          (torch.randn(32, 32, generator=g).mul_(0.2).to(dev),) + (torch.ones(32, device=dev), torch.zeros(32, device=dev)),
          (torch.randn(32, 64, generator=g).mul_(0.2).to(dev),) + (torch.ones(64, device=dev), torch.zeros(64, device=dev))]
    x512 = torch.randn(b * 256, 512, generator=g).to(dev)          # FP level on 256 points per cloud: 4 launches per step
    W512 = torch.randn(512, 512, generator=g).mul_(0.05).to(dev)
    x96 = torch.randn(b * 1024 * 32, 96, generator=g).to(dev)     # SA level 2, last layer + max-pool: the step's most expensive MLP launch
    W96 = torch.randn(96, 128, generator=g).mul_(0.1).to(dev)
    (idx1, _), = ops.query_ball_point_msg([0.5], [32], xyz, new1)
    one = lambda c: (torch.ones(c, device=dev), torch.zeros(c, device=dev))
    # RPN level 2, second scale as one launch (sa_chain2): table (b,4096,64), [64,96,128], nsample 32
    tab = torch.randn(b, 4096, 64, generator=g).to(dev)
    wx = torch.randn(3, 64, generator=g).mul_(0.3).to(dev)
    L2 = [(None,) + one(64), (torch.randn(64, 96, generator=g).mul_(0.1).to(dev),) + one(96),
          (torch.randn(96, 128, generator=g).mul_(0.1).to(dev),) + one(128)]
    L2[0] = (wx,) + one(64)
    f128 = torch.randn(b, 4096, 128, generator=g).to(dev)
    u4 = torch.randn(4, 128, generator=g).mul_(0.1).to(dev)
    t128 = torch.zeros(128, device=dev)
    torch.cuda.synchronize()
    work = {
        "group_point_kernel<4>": b * 1024 * 32 * (4 + 8 * 96),
        "three_interpolate_kernel<4>": b * 16384 * (24 + 16 * 256),
        "three_nn_interpolate_kernel<4>": b * (12 * 16384 + 12 * 4096) + b * 16384 * 16 * 256,
        "mlp_kernel<2, 2, 2, 2, false, false> rows262144 K260 N128": 4 * (b * 16384 * 260 + 260 * 128 + b * 16384 * 128),
        "mlp_kernel<2, 2, 2, 2, false, false> rows65536 K608 N256": 4 * (b * 4096 * 608 + 608 * 256 + b * 4096 * 256),
        "mlp_kernel<4, 1, 1, 2, false, false, 32> rows4096 K512 N512": 4 * (b * 256 * 512 + 512 * 512 + b * 256 * 512),
        "mlp_kernel<2, 2, 2, 2, false, false> rows524288 K96 N128 pool32": 4 * (b * 1024 * 32 * 96 + 96 * 128 + b * 1024 * 128),
        "ns_search_kernel<4, true> (culled three_nn + interpolate + affine epilogue) n16384 m4096 C128 c1=4": b * (12 * 16384 + 12 * 4096) + b * 16384 * (16 * 128 + 16),
        "bq_culled_kernel<2> n16384 m4096 r0.1/0.5 ns16/32": b * (12 * 16384 + 12 * 4096 + 4 * 4096 * 48 + 8 * 4096),
        "sa_chain_kernel<4, 32, 32, 64> rows2097152": b * 4096 * 32 * (4 + 16) + 4 * b * 4096 * 64,
        "mlp_kernel<4, 1, 1, 2, true, false>": b * 1024 * 32 * (4 + 4 * 99) + 4 * b * 1024 * 32 * 64,
        "sa_chain2_kernel<96, 6> rows524288 N64-96-128": b * 1024 * 32 * (4 + 12 + 4 * 64) + 4 * b * 1024 * 128,
    }
    for _ in range(REPS):
        ops.group_point(f1, idx2)
        ops.three_interpolate(f256, i3, w)
        ops.three_nn_interpolate_affine(xyz, new1, f128, pc, u4, t128)     # last RPN FP level as the stack runs it
        ops.mlp_layer(x260, W260, *one(128))
        ops.mlp_layer(x608, W608, *one(256))
        ops.mlp_layer(x96, W96, *one(128), pool=32)
        ops.mlp_layer(x512, W512, *one(512))
        ops.sa_group_mlp3_pool(xyz, feats, new1, idx1, W4)
        ops.sa_group_add_mlp2_pool(new1, new2, idx2, tab, 0, wx, L2)
        ops.sa_group_mlp_layer(new1, f1, new2, idx2, W99, *one(64))
        ops.query_ball_point_msg([0.1, 0.5], [16, 32], xyz, new1)
        ops.farthest_point_sample(4096, xyz)
        ops.farthest_point_sample(4096, xyz, shared=True)
    torch.cuda.synchronize()
    import json
    print(json.dumps({"algorithmic_bytes_per_launch": work}))


if __name__ == "__main__":
    main()
