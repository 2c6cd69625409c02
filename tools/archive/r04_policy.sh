#!/bin/bash
# round 4: forward policy -- tests on the product build + A/B timing of the three policies (rotated inputs)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_forward_policy.py tests/test_gpu_profiles.py -m gpu -x -q 2>&1 | tail -15
for rep in 1 2; do for sg in 1.0 2.0 3.0 4.0; do for pol in patch window adaptive; do
SEMIDETR_EXPERIMENTS=0 timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --iters 24 --sigma $sg --cold 6 --policy $pol --print-kernels 2>&1 | tail -2 | tr '\n' ' ' | sed "s/^/[sigma $sg $pol] /"; echo
done; done; done
