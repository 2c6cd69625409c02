#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_msda.py tests/test_gpu_fused.py tests/test_gpu_module.py -m gpu -x -q 2>&1 | tail -4
for v in 8 808 0; do timeout 120 python tools/msda_probe.py --shape micro --bs 2 --dir bwd --variant $v --iters 200 2>&1 | tail -1; done
for v in 32 0; do timeout 120 python tools/msda_probe.py --shape dec --bs 4 --lq 1100 --dir bwd --variant $v --iters 50 2>&1 | tail -1; done
timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 0 --iters 20 2>&1 | tail -1
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02_bench3.json 2> gpurun_out/r02_bench3.err; tail -2 gpurun_out/r02_bench3.err
python - <<PY
import json
d=json.load(open('gpurun_out/r02_bench3.json'))
print(d['value'], d['ms_per_step'], d['roofline']['traffic'])
print({k:v for k,v in d['microbench'].items() if k!='secondary_shapes'})
print(d['breakdown_ms_per_step'])
PY
