#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
SEMIDETR_EXPERIMENTS=1 SEMIDETR_TEST_VARIANT=0,910 timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "decoder" 2>&1 | tail -3
SEMIDETR_EXPERIMENTS=1 timeout 1200 python -m pytest tests/test_gpu_msda.py -q -m gpu -k "variants" 2>&1 | tail -3
bash tools/r03_quick.sh
