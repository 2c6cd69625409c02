#!/bin/bash
# bs 4: where do the patch kernel and the 768-thread window kernel cross?  rotated inputs, product library
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for sg in 3.5 4.0 4.5 5.0 6.0; do for pol in patch window; do
SEMIDETR_EXPERIMENTS=0 timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --iters 24 --sigma $sg --cold 6 --policy $pol 2>&1 | tail -1 | sed "s/^/[bs4 sigma $sg $pol] /"
done; done; done
