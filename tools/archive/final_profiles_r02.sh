#!/bin/bash
# Round-2 refresh on the GPU box: full GPU test suite, default bench line, traffic JSON, rocprofv3 kernel stats of the
# bench command (the summary that profiles/r02_bench_kernel_stats.txt is made from).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout -k 5 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r02_final_tests.log
cat gpurun_out/r02_final_tests.log
timeout -k 5 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 5 1200 python tools/measure_traffic.py > gpurun_out/r02_traffic.log 2>&1; tail -3 gpurun_out/r02_traffic.log
cp gpurun_out/r02_pmc_traffic.json profiles/r02_pmc_traffic.json
timeout -k 5 900 python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
tail -2 gpurun_out/r02_bench_default.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r02_prof_bench
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02_prof_bench -- python $R/bench.py --steps 10 --warmup 3 --no-micro --no-cpu-baseline > $R/gpurun_out/r02_prof_bench.json 2> $R/gpurun_out/r02_prof_bench.err
cd $R
python tools/summarize_prof.py stats gpurun_out/r02_prof_bench > gpurun_out/r02_bench_kernel_stats.txt
python tools/summarize_prof.py bygrid gpurun_out/r02_prof_bench > gpurun_out/r02_bench_kernel_by_grid.txt
head -30 gpurun_out/r02_bench_kernel_stats.txt
