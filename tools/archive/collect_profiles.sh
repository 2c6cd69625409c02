#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel-trace stats of the default bench command + PMC passes (FETCH_SIZE and
# WRITE_SIZE in separate runs, with --kernel-trace only) for the encoder-shape kernels.  Outputs under
# gpurun_out/; condense with tools/summarize_prof.py and commit the summaries under profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_bench_stats -- \
    python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_stdout.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_${C} -- \
      python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --iters 3 > /dev/null 2>&1
  timeout -k 5 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_pmc_micro_${C} -- \
      python $R/tools/msda_probe.py --shape micro --bs 2 --dir both --iters 3 > /dev/null 2>&1
done
tail -1 $R/gpurun_out/${TAG}_bench_stdout.txt | cut -c1-300
