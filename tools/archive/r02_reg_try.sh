#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention" 2>&1 | tail -3
SEMIDETR_TEST_VARIANT=0,69 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "encoder" 2>&1 | tail -3
for v in 65 69 690; do for bs in 4 1; do timeout 120 python tools/msda_probe.py --shape enc --bs $bs --dir bwd --variant $v --iters 20 2>&1 | tail -1; done; done
for s in 1.0 4.0; do timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 69 --iters 10 --sigma $s 2>&1 | tail -1; timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 65 --iters 10 --sigma $s 2>&1 | tail -1; done
