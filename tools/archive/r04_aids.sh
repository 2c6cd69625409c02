#!/bin/bash
# round 4: timing aids of the fused encoder backward (6902.. = parts switched off, results wrong) + SQ counters
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in ${VARS:-6900 6902 6903 6904 6905 6906 6907 6908 69}; do timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 20 2>&1 | tail -1; done
done
if [ -n "$PMC" ]; then
bash tools/pmc_sq.sh --shape enc --bs 4 --dir bwd --variant ${PMCVAR:-6900} --iters 3 2>&1 | grep -v "^$" | tail -40
fi
