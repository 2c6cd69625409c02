#!/bin/bash
# small kernels of the step (matcher, NMS, EMA, pseudo labels) after a change: their parity tests, then their kernel
# times inside the bench step.   usage: r02_small_kernels.sh [pytest files ...]
cd $GRAFT_REPO_ROOT
TESTS=${@:-tests/test_gpu_matcher.py tests/test_gpu_targets.py tests/test_gpu_nms.py tests/test_gpu_ema_pseudo.py tests/test_gpu_o2m.py}
timeout 900 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/small_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/small_stats -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/small_bench.log 2>&1
python - <<PY
import csv, glob, json
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/small_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if any(k in n for k in ("nms", "lsap", "pseudo", "match_cost", "ema_", "build_targets", "transform_bboxes", "o2m", "tal_")):
            print("%-40s calls %4s avg %7.1f us" % (n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40], r["Calls"], float(r["AverageNs"]) / 1e3))
for line in open("$GRAFT_REPO_ROOT/gpurun_out/small_bench.log"):
    if line.startswith("{"):
        d = json.loads(line); b = d["breakdown_ms_per_step"]
        print("step ms %.3f  hungarian %.3f  pseudo_label %.3f  ema %.3f" % (d["ms_per_step"], b["hungarian_batch"], b["pseudo_label"], b["ema"]))
PY
