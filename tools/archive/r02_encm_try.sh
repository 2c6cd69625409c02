#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fused.py -m gpu -x -q 2>&1 | tail -3
for v in 65 0; do for bs in 4 1; do timeout 120 python tools/msda_probe.py --shape enc --bs $bs --dir bwd --variant $v --iters 20 2>&1 | tail -1; done; done
