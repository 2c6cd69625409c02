#!/bin/bash
# bs 4, sigma 1 .. 3 px: patch kernel against the 768-thread window kernel, rotated inputs, product library (completes r04_rw_cross2.sh)
cd $GRAFT_REPO_ROOT
for sg in 1.0 1.5 2.0 2.5 3.0; do for pol in patch window; do
SEMIDETR_EXPERIMENTS=0 timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --iters 24 --sigma $sg --cold 6 --policy $pol 2>&1 | tail -1 | sed "s/^/[bs4 sigma $sg $pol] /"
done; done
