#!/bin/bash
# round 4: encoder backward at bs 1 (the supervised image): region shapes of the scatter, rotated inputs
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 69 692 693 691 694 698; do
timeout 120 python tools/msda_probe.py --shape enc --bs 1 --dir bwd --variant $v --iters 40 --cold 6 2>&1 | tail -1 | sed "s/^/[bs1 bwd $v] /"
done; done
bash tools/ab_variants.sh "69 693" --shape enc --bs 1 --dir bwd --iters 24 --cold 6 2>&1 | grep -v "^$" | tail -12
