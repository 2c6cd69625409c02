#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 70 72; do timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 20 2>&1 | tail -1; done
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02_dest_prof2 -o dest --output-format csv -- python $GRAFT_REPO_ROOT/tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 70 --iters 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r02_dest_prof2 -name "*kernel_stats*" | head -1 | xargs -I{} sh -c 'cut -c1-200 {} | head -8'
