#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_msda.py tests/test_gpu_fullsize.py tests/test_gpu_fused.py tests/test_gpu_module.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-micro > gpurun_out/r02_bench4.json 2> gpurun_out/r02_bench4.err; tail -2 gpurun_out/r02_bench4.err
python - <<PY
import json
d=json.load(open('gpurun_out/r02_bench4.json'))
print(d['value'], d['ms_per_step'])
print(d['breakdown_ms_per_step'])
PY
