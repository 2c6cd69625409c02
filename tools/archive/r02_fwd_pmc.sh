#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 500; do
  echo "=== fwd variant $v"
  bash tools/pmc_fwd.sh --shape enc --bs 4 --dir fwd --variant $v --iters 5 2>&1 | grep -v "^$" | tail -12
  bash tools/pmc_sq.sh --shape enc --bs 4 --dir fwd --variant $v --iters 5 2>&1 | grep -v "^$" | tail -18
done > gpurun_out/r02_fwd_res_pmc.txt 2>&1
cat gpurun_out/r02_fwd_res_pmc.txt
