#!/bin/bash
# like r04_rw_waves.sh with the sigmas as a second argument
cd $GRAFT_REPO_ROOT
VARS=${1:-"743 748"}; SIGS=${2:-"3.0 4.0 5.0"}
for rep in 1 2; do for sg in $SIGS; do for v in $VARS; do
timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --fvariant $v --iters 24 --sigma $sg --cold 6 --policy patch 2>&1 | tail -1 | sed "s/^/[sigma $sg fwd $v] /"
done; done; done
