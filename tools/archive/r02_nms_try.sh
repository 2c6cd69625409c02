#!/bin/bash
# NMS class kernel after a change: parity tests, then its kernel time inside the bench step
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_nms.py tests/test_gpu_ema_pseudo.py -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/nms_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/nms_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/nms_bench.log 2>&1
python - <<PY
import csv, glob, json
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/nms_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nms" in r["Name"] or "lsap" in r["Name"] or "pseudo" in r["Name"]:
            print("%-60s calls %s avg %.1f us" % (r["Name"].split("(")[0][-58:], r["Calls"], float(r["AverageNs"]) / 1e3))
for line in open("$GRAFT_REPO_ROOT/gpurun_out/nms_bench.log"):
    if line.startswith("{"):
        d = json.loads(line); print("step ms", d["ms_per_step"], "pseudo_label", d["breakdown_ms_per_step"]["pseudo_label"])
PY
