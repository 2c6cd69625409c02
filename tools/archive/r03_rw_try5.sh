#!/bin/bash
# region-window forward after (a) two out-of-window samples per trip, (b) streaming output stores
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
SEMIDETR_EXPERIMENTS=1 timeout 900 python -m pytest tests/test_gpu_msda.py -x -q -m gpu -k "encoder_self_attention" 2>&1 | tail -2
for cfg in 0 2 5 6; do
    timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir fwd --fvariant $((700+cfg)) --variant 0 --iters 30 2>&1 | grep "us  alg" | sed "s/^/[cfg $cfg] /"
done
timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --iters 30 2>&1 | grep "us  alg" | sed "s/^/[product] /"
for sg in 1 4; do
  timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir fwd --fvariant 702 --variant 0 --sigma $sg --iters 20 2>&1 | grep "us  alg" | sed "s/^/[cfg 2 sigma $sg] /"
  timeout 300 python $R/tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --sigma $sg --iters 20 2>&1 | grep "us  alg" | sed "s/^/[product sigma $sg] /"
done
