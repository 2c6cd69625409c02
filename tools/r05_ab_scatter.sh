#!/bin/bash
# A/B old vs new product library: encoder backward bs 4 and bs 1 (kernel averages), then parity tests on the new one
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/ab_libs.sh --shape enc --bs 4 --dir bwd --variant 0 --iters 12 --cold 6 --settle 3 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_msda.py tests/test_gpu_fullsize.py tests/test_gpu_fused.py -x -q -m gpu 2>&1 | tail -4
