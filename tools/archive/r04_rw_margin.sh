#!/bin/bash
# round 4: the product's window shape with margin 5 on the coarse levels (715 / 716) against margin 4 (702) and the patch kernel (0)
cd $GRAFT_REPO_ROOT
SEMIDETR_EXPERIMENTS=1 timeout 600 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention and f702b" -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do for sg in 1.0 2.0 2.5 3.0 4.0; do for v in 702 715 716; do
timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --fvariant $v --iters 24 --sigma $sg --cold 6 --policy patch 2>&1 | tail -1 | sed "s/^/[sigma $sg fwd $v] /"
done; done; done

