#!/bin/bash
# quick check on the product library: MSDA parity tests that need no variants + the default bench step (no extras)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_msda.py tests/test_gpu_fullsize.py tests/test_gpu_fused.py tests/test_gpu_module.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-micro --no-cpu-baseline > $O/r03_q.json 2> $O/r03_q.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_q.json").read().strip().splitlines()[-1])
print({k: round(d[k], 2) for k in ("value", "ms_per_step")}, {k: round(v, 2) for k, v in d["breakdown_ms_per_step"].items() if "enc" in k})
PY
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-micro --no-cpu-baseline --io raw > $O/r03_q.json 2> $O/r03_q.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_q.json").read().strip().splitlines()[-1])
print("raw", {k: round(d[k], 2) for k in ("value", "ms_per_step")}, {k: round(v, 2) for k, v in d["breakdown_ms_per_step"].items() if "enc" in k})
PY
