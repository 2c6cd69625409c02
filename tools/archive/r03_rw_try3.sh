#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_msda.py -x -q -k "encoder_self_attention" > $O/r03_rw_small.log 2>&1
tail -3 $O/r03_rw_small.log
SEMIDETR_TEST_VARIANT=700,7000 timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "encoder" > $O/r03_rw_full.log 2>&1
tail -3 $O/r03_rw_full.log
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/tools/r03_rw_dbg.py 2>&1 | tail -26
for cfg in 0 1 2 3 4 8 9; do
    timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --fvariant $((700+cfg)) --variant $((7000+cfg)) --iters 20 2>&1 | grep "us  alg" | sed "s/^/[cfg $cfg] /"
done
timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --variant 0 --iters 20 2>&1 | grep "us  alg" | sed "s/^/[patch] /"
for sg in 1 4; do
  timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --fvariant 700 --variant 7000 --sigma $sg --iters 20 2>&1 | grep "us  alg" | sed "s/^/[cfg 0 sigma $sg] /"
done
rm -rf $O/r03_rw_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_rw_prof -- python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --fvariant 700 --variant 7000 --iters 12 > $O/r03_rw_prof.log 2>&1
python $R/tools/summarize_prof.py stats $O/r03_rw_prof | head -8
