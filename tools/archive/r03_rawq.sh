#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-micro --no-cpu-baseline --io raw > $O/r03_q.json 2> $O/r03_q.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03_q.json").read().strip().splitlines()[-1])
print("raw", {k: round(d[k], 2) for k in ("value", "ms_per_step")}, {k: round(v, 2) for k, v in d["breakdown_ms_per_step"].items() if "enc" in k})
PY
done
