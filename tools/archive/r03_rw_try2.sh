#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_msda.py -x -q -k "encoder_self_attention" > $O/r03_rw_small.log 2>&1
tail -5 $O/r03_rw_small.log
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/tools/r03_rw_dbg.py 2>&1 | tail -24
for cfg in 0 8 9; do
    timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --fvariant $((700+cfg)) --variant $((7000+cfg)) --iters 20 2>&1 | grep "us  alg" | sed "s/^/[cfg $cfg] /"
done
