#!/bin/bash
# round 4: the region-window GATHER (all-level windows, 8 x 16 regions) with leaner schedules, rotated inputs, against the patch gather
cd $GRAFT_REPO_ROOT
bash tools/ab_variants.sh "0 7000 7001 7002 7003" --shape enc --bs 4 --dir bwd --iters 12 --cold 6 2>&1 | grep "avg " | grep -v "fill\|scatter" | sort | uniq | cut -c1-110
