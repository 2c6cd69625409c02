#!/bin/bash
# patch vs window forward by sample spread (product library), far share printed by the policy state
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for sg in 3.0 4.0 5.0 6.0 7.0 8.0; do
for pol in patch window; do
SEMIDETR_EXPERIMENTS=0 timeout 300 python tools/msda_probe.py --shape enc --bs ${BS:-4} --dir fwd --iters 40 --cold 6 --variant 0 --policy $pol --sigma $sg 2>&1 | grep "us  alg" | awk -v p=$pol -v s=$sg '{print "sigma", s, p, $6, "us"}'
done
done
