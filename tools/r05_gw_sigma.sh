#!/bin/bash
# encoder backward (bs 4 / bs 1) by sample spread: patch gather vs lane-per-sample window gather
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for bs in 4 1; do for s in 1 2 3 4 5 6 8; do for pol in patch window; do
  echo "bs $bs sigma $s $pol: $(SEMIDETR_EXPERIMENTS=0 timeout 120 python tools/msda_probe.py --shape enc --bs $bs --dir bwd --iters 12 --cold 6 --variant 0 --policy $pol --sigma $s 2>&1 | grep 'us  alg' | awk '{print $6, $7}')"
done; done; done
