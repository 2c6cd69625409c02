#!/bin/bash
# full GPU suite on the product library (+ the variant tests on the experiments build through tests/test_gpu_experiments.py),
# then a short bench run
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/r03_suite.log 2>&1
tail -8 $O/r03_suite.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/r03_bench_quick.json 2> $O/r03_bench_quick.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_bench_quick.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step")}, d.get("roofline", {}).get("frac"))
PY
tail -3 $O/r03_bench_quick.err
