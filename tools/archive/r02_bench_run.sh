#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r02_bench2.json 2> gpurun_out/r02_bench2.err; echo "bench rc=$?"; tail -5 gpurun_out/r02_bench2.err
timeout 1200 python tools/measure_traffic.py > gpurun_out/r02_traffic.log 2>&1; echo "traffic rc=$?"; tail -12 gpurun_out/r02_traffic.log
