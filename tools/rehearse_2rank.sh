#!/bin/bash
# Two ranks sharing ONE GPU with a gloo group: rehearsal of the N>1 path of bench.py on a 1-GPU box
# (RCCL needs one device per rank; the product path on a real node is backend nccl).
cd $GRAFT_REPO_ROOT
export PRCNN_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 8 --warmup 4 --no-cpu-baseline --no-extras --dump-records gpurun_out/rec_w2.npy \
    > gpurun_out/bench_w2.json 2> gpurun_out/bench_w2.err
echo "rc=$?"
tail -c 600 gpurun_out/bench_w2.json
