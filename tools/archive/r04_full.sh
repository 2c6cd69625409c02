#!/bin/bash
# round 4: traffic + TA evidence -> profiles/, the whole GPU suite, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/measure_traffic.py > gpurun_out/r04_traffic.log 2>&1; tail -3 gpurun_out/r04_traffic.log
cp gpurun_out/r04_pmc_traffic.json profiles/ 2>/dev/null
timeout 900 bash tools/r04_fwd_ta_evidence.sh > gpurun_out/r04_ta.log 2>&1; tail -3 gpurun_out/r04_ta.log
cp gpurun_out/r04_fwd_enc_TA.json gpurun_out/r04_fwd_enc_TA.txt profiles/ 2>/dev/null
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
timeout 900 python bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.err; tail -3 gpurun_out/r04_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_bench_line.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")})
print({k: round(v, 3) for k, v in d["breakdown_ms_per_step"].items()})
print("roofline", {k: d["roofline"][k] for k in ("kernel", "frac", "avg_launch_us", "traffic")})
print("l1", {k: d["roofline_l1"][k] for k in ("frac", "frac_at_observed_clock", "observed_clock_mhz", "ta_busy_frac_pmc")})
print({k: v for k, v in d.items() if k.startswith("microbench_cold")})
print(json.dumps(d.get("flavours"), indent=0)[:1500])
print("cpu", {k: d["cpu_baseline"][k] for k in ("value", "cores", "enc_fwd_s", "enc_bwd_s", "min_max_s")})
PY
