#!/bin/bash
cd $GRAFT_REPO_ROOT
SEMIDETR_EXPERIMENTS=1 timeout 900 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention and (b6900 or b6909 or b6983 or b6984)" -p no:cacheprovider 2>&1 | tail -3
VARS="69 6983 6984" bash tools/r04_aids.sh
bash tools/ab_variants.sh "69 6983" --shape enc --bs 4 --dir bwd --iters 12 2>&1 | grep -v "^$" | tail -20
