#!/bin/bash
# round 4: encoder backward variants re-measured with ROTATED inputs (they were ranked with replayed inputs in rounds 2-3)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
bash tools/ab_variants.sh "0 6962 6952 922 698 6981" --shape enc --bs 4 --dir bwd --iters 12 --cold 6 2>&1 | grep "avg " | sort | uniq | head -40
done
