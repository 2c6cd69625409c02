#!/bin/bash
# decoder forward: rows per workgroup (split 1 / 2 / 4 = 32 / 16 / 8 query rows per 256-thread workgroup), experiments library
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for bs in 4 1; do for lq in 1100 900 300; do for v in 1 2 4; do
SEMIDETR_EXPERIMENTS=1 timeout 300 python tools/msda_probe.py --shape dec --lq $lq --bs $bs --dir fwd --iters 100 --cold 6 --fvariant $v --variant 0 2>&1 | grep "us  alg" | awk -v b=$bs -v l=$lq -v v=$v '{print "bs", b, "Lq", l, "split", v, $6, "us"}'
done; done; done
