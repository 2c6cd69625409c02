#!/bin/bash
# Control-flow check of the N > 1 bench path on a one-GPU box: 2 ranks share cuda:0, gloo instead of RCCL.
export SEMIDETR_BENCH_SHARE_GPU=1 SEMIDETR_DIST_BACKEND=gloo
timeout -k 5 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600
