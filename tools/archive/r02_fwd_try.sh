#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention" 2>&1 | tail -3
for v in 0 500 501 502 503 504 505; do timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant $v --iters 30 2>&1 | tail -1; done
