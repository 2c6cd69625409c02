#!/bin/bash
# round 4: region-window forward configurations 700..706 with ROTATED inputs (they were chosen with replayed inputs in round 3)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for sg in 2.0 1.0; do for v in 0 700 701 702 703 704 705 706; do
timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --fvariant $v --iters 24 --sigma $sg --cold 6 --policy patch 2>&1 | tail -1 | sed "s/^/[sigma $sg fwd $v] /"
done; done; done
