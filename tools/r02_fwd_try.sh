#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention" 2>&1 | tail -6
for v in 0 500; do for bs in 4 1; do timeout 120 python tools/msda_probe.py --shape enc --bs $bs --dir fwd --variant $v --iters 30 2>&1 | tail -1; done; done
