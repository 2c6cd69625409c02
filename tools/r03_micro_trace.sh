#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for PART in 0 1 2; do
for FS in 1 4; do
export SEMIDETR_PART=$PART SEMIDETR_FSPLIT=$FS
rm -rf $O/mt; rocprofv3 --kernel-trace --stats -d $O/mt --output-format csv -- python $R/tools/msda_probe.py --shape micro --bs 2 --dir bwd --cold 8 --iters 160 > $O/mt.log 2>&1
f=$(find $O/mt -name "*kernel_stats.csv" | head -1); echo "part $PART fsplit $FS: $(grep lvl_merged $f | awk -F, '{print $(NF-4), $(NF-2), $(NF-1)}')"
done; done
