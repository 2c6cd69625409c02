#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/measure_traffic.py > gpurun_out/r04_traffic.log 2>&1; tail -4 gpurun_out/r04_traffic.log | cut -c1-300
cp gpurun_out/r04_pmc_traffic.json profiles/ 2>/dev/null
timeout 600 python -m pytest tests/test_gpu_forward_policy.py tests/test_gpu_profiles.py -m gpu -x -q 2>&1 | tail -3
