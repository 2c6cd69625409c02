#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export SEMIDETR_EXPERIMENTS=0
for WS in 0 2 3 4 5 6 8; do
export SEMIDETR_WS=$WS
echo "ws $WS: $(python tools/msda_probe.py --shape enc --bs 4 --dir fwd --iters 30 | tail -1)  | $(python tools/msda_probe.py --shape enc --bs 1 --dir fwd --iters 30 | tail -1 | cut -c30-50)"
done
SEMIDETR_WS=4 timeout 900 python -m pytest tests/test_gpu_msda.py tests/test_gpu_fullsize.py tests/test_gpu_fused.py -q -m gpu 2>&1 | tail -3
