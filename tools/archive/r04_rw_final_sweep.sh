#!/bin/bash
# round 4: the product's two forward kernels over the sample spread, rotated inputs (718 = the product's window configuration)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for sg in 1.0 1.5 2.0 2.5 3.0 3.5 4.0; do for v in 0 718; do
timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --fvariant $v --iters 24 --sigma $sg --cold 6 --policy patch 2>&1 | tail -1 | sed "s/^/[sigma $sg fwd $v] /"
done; done; done
