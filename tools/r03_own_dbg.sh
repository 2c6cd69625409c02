#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
export SEMIDETR_EXPERIMENTS=0
for D in 10 16 14 15 13 11 12; do
export SEMIDETR_OLDDBG=$D
echo "old dbg $D: $(python tools/msda_probe.py --shape micro --bs 2 --dir bwd --cold 8 --iters 160 | tail -1 | cut -c1-50)  $(python tools/msda_probe.py --shape dec --bs 4 --lq 1100 --dir bwd --iters 50 | tail -1 | cut -c1-50)"
done
