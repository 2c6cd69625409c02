#!/bin/bash
# round 4: encoder forward, patch kernel (0) vs region-window LDS forward (702 / 700), inputs replayed (--cold 1) and rotated (--cold 6)
cd $GRAFT_REPO_ROOT
for cold in 1 6; do for sg in 1.0 2.0 4.0; do for v in 0 702 700; do
timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --fvariant $v --iters 24 --sigma $sg --cold $cold 2>&1 | tail -1 | sed "s/^/[cold $cold sigma $sg fwd $v] /"
done; done; done
for cold in 1 6; do for v in 0 7002; do
timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 12 --cold $cold 2>&1 | tail -1 | sed "s/^/[cold $cold bwd $v] /"
done; done
