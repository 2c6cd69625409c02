#!/bin/bash
# Round-3 refresh on the GPU box: full GPU test suite (product library; variant tests on the experiments build through
# tests/test_gpu_experiments.py), smoke, traffic JSON (rocprofv3 --pmc passes), the bench line in its four flavours, and the
# rocprofv3 kernel statistics of the default bench command (-> profiles/r03_bench_kernel_stats.txt / _by_grid.txt).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout -k 5 1500 python tools/measure_traffic.py > gpurun_out/r03_traffic.log 2>&1; tail -3 gpurun_out/r03_traffic.log
[ -f gpurun_out/r03_pmc_traffic.json ] && cp gpurun_out/r03_pmc_traffic.json profiles/r03_pmc_traffic.json
timeout -k 5 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r03_final_tests.log
cat gpurun_out/r03_final_tests.log
timeout -k 5 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 5 900 python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
tail -2 gpurun_out/r03_bench_default.err
timeout 600 python bench.py --steps 10 --warmup 3 --io raw --no-cpu-baseline > gpurun_out/r03_bench_raw.json 2> gpurun_out/r03_bench_raw.err
timeout 600 python bench.py --steps 10 --warmup 3 --recipe full --no-cpu-baseline > gpurun_out/r03_bench_full.json 2> gpurun_out/r03_bench_full.err
timeout 600 python bench.py --steps 10 --warmup 3 --recipe full --io raw --no-cpu-baseline > gpurun_out/r03_bench_full_raw.json 2> gpurun_out/r03_bench_full_raw.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r03_prof_bench
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof_bench -- python $R/bench.py --steps 10 --warmup 3 --no-micro --no-cpu-baseline > $R/gpurun_out/r03_prof_bench.json 2> $R/gpurun_out/r03_prof_bench.err
cd $R
python tools/summarize_prof.py stats gpurun_out/r03_prof_bench > gpurun_out/r03_bench_kernel_stats.txt
python tools/summarize_prof.py bygrid gpurun_out/r03_prof_bench > gpurun_out/r03_bench_kernel_by_grid.txt
head -24 gpurun_out/r03_bench_kernel_by_grid.txt
python - <<'PY'
import json
for n in ("default", "raw", "full", "full_raw"):
    try:
        d = json.loads(open(f"gpurun_out/r03_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, {k: round(d[k], 2) for k in ("value", "ms_per_step")}, "frac", round(d["roofline"]["frac"], 4), d["roofline"]["kernel"], "traffic", d["roofline"]["traffic"])
    except Exception as e:
        print(n, "FAILED", e)
PY
