#!/bin/bash
mkdir -p gpurun_out
timeout -k 5 600 python -m pytest tests/test_gpu_nms.py -m gpu -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_nms -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 > /dev/null 2>&1
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/prof_nms/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nms" in r["Name"] or "transform" in r["Name"] or "pseudo" in r["Name"]:
            print(r["Name"][:60], r["Calls"], r["AverageNs"])
PY
