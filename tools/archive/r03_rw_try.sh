#!/bin/bash
# round 3: region-window forward / gather (msda_rw.h) -- parity on the small and the full-size shapes, then per-kernel
# times of every configuration against the patch kernels on ONE box.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_msda.py -x -q -k "encoder_self_attention and (700 or 701 or 702 or 703 or 704)" > $O/r03_rw_small.log 2>&1
tail -5 $O/r03_rw_small.log
SEMIDETR_TEST_VARIANT=700,7000 timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "encoder" > $O/r03_rw_full.log 2>&1
tail -5 $O/r03_rw_full.log
cd /tmp && export TMPDIR=/tmp
for cfg in 0 1 2 3 4; do
  for sg in 2; do
    timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --fvariant $((700+cfg)) --variant $((7000+cfg)) --sigma $sg --iters 20 2>&1 | grep "us  alg" | sed "s/^/[cfg $cfg sigma $sg] /"
  done
done
timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --variant 0 --sigma 2 --iters 20 2>&1 | grep "us  alg" | sed "s/^/[patch sigma 2] /"
for sg in 1 4; do
  timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --fvariant 700 --variant 7000 --sigma $sg --iters 20 2>&1 | grep "us  alg" | sed "s/^/[cfg 0 sigma $sg] /"
  timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --variant 0 --sigma $sg --iters 20 2>&1 | grep "us  alg" | sed "s/^/[patch sigma $sg] /"
done
timeout 300 python $R/tools/msda_probe.py --shape enc --bs 1 --dir both --fvariant 700 --variant 7000 --iters 20 2>&1 | grep "us  alg" | sed "s/^/[cfg 0 bs1] /"
timeout 300 python $R/tools/msda_probe.py --shape enc --bs 1 --dir both --variant 0 --iters 20 2>&1 | grep "us  alg" | sed "s/^/[patch bs1] /"
# per-kernel split of the backward (gather vs scatter) for the default configuration
rm -rf $O/r03_rw_prof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_rw_prof -- python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --fvariant 700 --variant 7000 --iters 12 > $O/r03_rw_prof.log 2>&1
python $R/tools/summarize_prof.py stats $O/r03_rw_prof | head -12
