#!/bin/bash
# zero fill of grad_value folded into the gather launch (variants 901: strips, 6990: encoder) against the defaults
cd $GRAFT_REPO_ROOT
SEMIDETR_TEST_VARIANT=0,6990 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "encoder" 2>&1 | tail -1
SEMIDETR_TEST_VARIANT=0,901 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "decoder or five" 2>&1 | tail -1
for rep in 1 2; do
for v in 0 901; do
  timeout 120 python tools/msda_probe.py --shape micro --bs 2 --dir bwd --variant $v --iters 300 2>&1 | tail -1
  timeout 120 python tools/msda_probe.py --shape dec --bs 1 --dir bwd --variant $v --iters 100 2>&1 | tail -1
  timeout 120 python tools/msda_probe.py --shape dec --bs 4 --dir bwd --variant $v --iters 100 2>&1 | tail -1
done
for v in 0 6990; do
  timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 20 2>&1 | tail -1
  timeout 120 python tools/msda_probe.py --shape enc --bs 1 --dir bwd --variant $v --iters 40 2>&1 | tail -1
done
done
