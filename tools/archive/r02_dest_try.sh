#!/bin/bash
# parity of the self-attention backward + timing of the destination-owned kernel vs the windowed one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fused.py tests/test_gpu_msda.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r02_dest_tests.log
for v in 65 70 71; do
  for bs in 4 1; do
    timeout 120 python tools/msda_probe.py --shape enc --bs $bs --dir bwd --variant $v --iters 20 2>&1 | tail -1
  done
done > gpurun_out/r02_dest_time.log 2>&1
for s in 1.0 4.0; do timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 70 --iters 10 --sigma $s 2>&1 | tail -1; timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 65 --iters 10 --sigma $s 2>&1 | tail -1; done >> gpurun_out/r02_dest_time.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02_dest_prof -o dest -- python $GRAFT_REPO_ROOT/tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 70 --iters 10 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r02_dest_prof -name "*kernel_stats*" | head -1 | xargs -I{} sh -c 'head -8 {}' > gpurun_out/r02_dest_kstats.txt
cat gpurun_out/r02_dest_tests.log gpurun_out/r02_dest_time.log gpurun_out/r02_dest_kstats.txt
