#!/bin/bash
# Round 6, first GPU call: GPU suite, the new default bench line (fused prologue + mask), rocprofv3 kernel trace of the headline step.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R && mkdir -p gpurun_out
timeout -k 5 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r06_first_tests.log
cat gpurun_out/r06_first_tests.log
timeout -k 5 900 python bench.py > gpurun_out/r06_bench_first.json 2> gpurun_out/r06_bench_first.err
tail -2 gpurun_out/r06_bench_first.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r06_prof_first
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_prof_first -- python $R/bench.py --steps 10 --warmup 3 --no-micro --no-flavours --no-cpu-baseline > $R/gpurun_out/r06_prof_first.json 2> $R/gpurun_out/r06_prof_first.err
cd $R
python tools/summarize_prof.py bygrid gpurun_out/r06_prof_first > gpurun_out/r06_first_by_grid.txt
head -30 gpurun_out/r06_first_by_grid.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_first.json").read().strip().splitlines()[-1])
print({k: round(d[k], 2) for k in ("value", "ms_per_step")}, d["config"]["io"], d["config"]["masked"], "frac", round(d["roofline"]["frac"], 4), d["roofline"]["kernel"], d["roofline"]["kernel_symbols"])
print({k: round(v, 3) for k, v in d["breakdown_ms_per_step"].items()})
print({k: (round(v["ms_per_step"], 2), round(v["images_per_s"], 1), round(v["dominant_avg_launch_us"],1), round(v["enc_bwd_avg_launch_us"],1)) for k, v in d["flavours"].items() if isinstance(v, dict)})
print({k: round(v, 3) for k, v in d.items() if k.startswith("microbench_cold") and isinstance(v, float)})
PY
rm -rf gpurun_out/r06_prof_first/*/*.db 2>/dev/null
