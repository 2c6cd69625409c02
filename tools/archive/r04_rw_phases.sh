#!/bin/bash
# round 4: where the product's region-window forward (512 threads, 16 x 16 regions, level 0 global, margin 4) spends its time
cd $GRAFT_REPO_ROOT
timeout 300 python tools/r03_rw_dbg.py 712 2>&1 | tail -12
for rep in 1 2; do for v in 702 713 714; do
timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --fvariant $v --iters 24 --sigma 2.0 --cold 6 --policy patch 2>&1 | tail -1 | sed "s/^/[fwd $v] /"
done; done
