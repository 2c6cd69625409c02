#!/bin/bash
timeout -k 5 600 python -m pytest tests/test_gpu_msda.py tests/test_gpu_fused.py tests/test_gpu_module.py -m gpu -x -q 2>&1 | tail -3
for v in 0 67 66; do
  timeout -k 5 120 python tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 30 2>&1 | grep -v amdgpu.ids | tail -1
  timeout -k 5 120 python tools/msda_probe.py --shape enc --bs 1 --dir bwd --variant $v --iters 30 2>&1 | grep -v amdgpu.ids | tail -1
done
cd /tmp && export TMPDIR=/tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_g -- python $GRAFT_REPO_ROOT/tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 0 --iters 10 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_g/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda" in r["Name"] or "fill" in r["Name"]:
            print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
