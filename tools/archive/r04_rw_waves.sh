#!/bin/bash
# round 4: does the region-window forward want more waves?  1024-thread workgroups (16 waves per CU) against 512 at equal margin,
# rotated inputs; each configuration's result is compared with the patch kernel's first
cd $GRAFT_REPO_ROOT
VARS=${1:-"730 731 732 733 718"}
for rep in 1 2; do for sg in 2.0 1.0; do for v in $VARS; do
timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --fvariant $v --iters 24 --sigma $sg --cold 6 --policy patch --check 2>&1 | tail -2 | sed "s/^/[sigma $sg fwd $v] /"
done; done; done
