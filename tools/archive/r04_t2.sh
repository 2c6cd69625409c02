#!/bin/bash
cd $GRAFT_REPO_ROOT
SEMIDETR_EXPERIMENTS=1 timeout 900 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention and (6900 or 6909)" -p no:cacheprovider 2>&1 | tail -3
SEMIDETR_EXPERIMENTS=1 SEMIDETR_TEST_VARIANT=0,6909 timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
VARS="6900 6909 6910 6911 69" bash tools/r04_aids.sh
