#!/bin/bash
# one image (616 regions x 8 heads for 256 CUs): patch against window kernel by sample spread, rotated inputs, product library
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for sg in 2.0 3.0 3.5 4.0; do for pol in patch window; do
SEMIDETR_EXPERIMENTS=0 timeout 120 python tools/msda_probe.py --shape enc --bs 1 --dir fwd --variant 0 --iters 40 --sigma $sg --cold 12 --policy $pol 2>&1 | tail -1 | sed "s/^/[bs1 sigma $sg $pol] /"
done; done; done
