#!/bin/bash
mkdir -p gpurun_out
timeout -k 5 900 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err; tail -3 gpurun_out/bench_quick.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_quick.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"])
print("warmup_stage", d.get("warmup_stage"))
print("micro", {k: d["microbench"][k] for k in d["microbench"] if "us" in k})
print("breakdown", {k: round(v, 3) for k, v in d["breakdown_ms_per_step"].items()})
PY
