#!/bin/bash
# gather half of the encoder backward after a change: parity (module + op + full size), then timing at bs 4 / bs 1
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_msda.py tests/test_gpu_module.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
for bs in 4 1; do timeout 120 python tools/msda_probe.py --shape enc --bs $bs --dir bwd --variant 0 --iters 20 2>&1 | tail -1; done
timeout 120 python tools/msda_probe.py --shape dec --bs 4 --dir bwd --variant 0 --iters 50 2>&1 | tail -1
timeout 120 python tools/msda_probe.py --shape micro --bs 2 --dir bwd --variant 0 --iters 200 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/gath_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gath_stats -- python $GRAFT_REPO_ROOT/tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 0 --iters 10 > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/gath_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"] or "fill" in r["Name"]:
            print("%-70s avg %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-68:], float(r["AverageNs"]) / 1e3))
PY
