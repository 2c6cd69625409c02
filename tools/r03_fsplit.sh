#!/bin/bash
# merged backward (arbitrary query sets): finer chunks for the levels too large to bucket (SEMIDETR_FSPLIT, temporary env switch)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for FS in 1 2 3 4 6; do
  export SEMIDETR_FSPLIT=$FS
  echo "== fsplit $FS"
  for i in 1 2; do python tools/msda_probe.py --shape micro --bs 2 --dir bwd --cold 8 --iters 160 | tail -1; done
  python tools/msda_probe.py --shape dec --bs 4 --lq 1100 --dir bwd --iters 50 | tail -1
  python tools/msda_probe.py --shape dec --bs 1 --lq 1100 --dir bwd --iters 50 | tail -1
done
export SEMIDETR_FSPLIT=3
timeout 900 python -m pytest tests/test_gpu_msda.py -q -m gpu -x 2>&1 | tail -2
