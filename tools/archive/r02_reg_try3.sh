#!/bin/bash
# three region-scatter workgroups per CU (variant 698: 176 queries per pass, 52 KB LDS, <= 80 VGPRs) against the default
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention" 2>&1 | tail -3
for v in 69 698 699; do for bs in 4 1; do timeout 120 python tools/msda_probe.py --shape enc --bs $bs --dir bwd --variant $v --iters 20 2>&1 | tail -1; done; done
for v in 69 698; do timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 10 --sigma 4.0 2>&1 | tail -1; done
