#!/bin/bash
# LDS-window forward (variant 600) against the plain patch kernel: parity, then timing
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention" 2>&1 | tail -2
SEMIDETR_TEST_VARIANT=600,0 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "encoder" 2>&1 | tail -1
for rep in 1 2; do for v in 0 600; do for bs in 4 1; do timeout 120 python tools/msda_probe.py --shape enc --bs $bs --dir fwd --fvariant $v --iters 30 --print-kernels 2>&1 | tail -2 | tr '\n' ' '; echo; done; done; done
for s in 1.0 4.0; do for v in 0 600; do timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --fvariant $v --iters 20 --sigma $s 2>&1 | tail -1 | sed "s/^/sigma $s: /"; done; done
