#!/bin/bash
# round 4: fused encoder backward (msda_bwd_enc_fused_d32 = product dispatch) against gather + region scatter (variant 69)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder or full_size or wide_level" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fused.py tests/test_gpu_module.py -m gpu -x -q 2>&1 | tail -5
for v in 0 69; do for bs in 4 1; do timeout 120 python tools/msda_probe.py --shape enc --bs $bs --dir bwd --variant $v --iters 20 --print-kernels 2>&1 | tail -2; done; done
for v in 0 69; do timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 10 --sigma 4.0 2>&1 | tail -1; done
for v in 0 69; do timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 10 --sigma 1.0 2>&1 | tail -1; done
timeout 120 python tools/r04_phases.py 6901 4 2>&1 | tail -12
timeout 120 python tools/r04_phases.py 696 4 2>&1 | tail -12
