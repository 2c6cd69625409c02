#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention" 2>&1 | tail -3
SEMIDETR_TEST_VARIANT=0,70 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
for v in 65 70 71 72; do timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 20 2>&1 | tail -1; done
