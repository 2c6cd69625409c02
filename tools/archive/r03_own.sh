#!/bin/bash
# merged backward (arbitrary query sets): parity + timing
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
export SEMIDETR_EXPERIMENTS=0
timeout 900 python -m pytest tests/test_gpu_msda.py tests/test_gpu_fused.py -q -m gpu 2>&1 | tail -3
for i in 1 2; do python tools/msda_probe.py --shape micro --bs 2 --dir bwd --cold 8 --iters 160 | tail -1; done
python tools/msda_probe.py --shape dec --bs 4 --lq 1100 --dir bwd --iters 50 | tail -1
python tools/msda_probe.py --shape dec --bs 1 --lq 1100 --dir bwd --iters 50 | tail -1
python tools/msda_probe.py --shape dec --bs 4 --lq 300 --dir bwd --iters 50 | tail -1
