#!/bin/bash
# Round-end refresh on the GPU box: default bench line, rocprofv3 kernel stats of the bench command, PMC passes.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout -k 5 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -2 gpurun_out/bench_default.err
./tools/collect_profiles.sh r01j
