#!/bin/bash
# round 4: the product's window configuration (718) at bs 1 and bs 2 against the patch kernel, rotated inputs
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for bs in 1 2; do for sg in 1.0 2.0 3.0; do for v in 0 718 719; do
timeout 120 python tools/msda_probe.py --shape enc --bs $bs --dir fwd --variant 0 --fvariant $v --iters 40 --sigma $sg --cold 6 --policy patch 2>&1 | tail -1 | sed "s/^/[bs$bs sigma $sg fwd $v] /"
done; done; done; done
