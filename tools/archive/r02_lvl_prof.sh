#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "micro --bs 2" "dec --bs 4 --lq 1100"; do
rm -rf $R/gpurun_out/lvlprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/lvlprof -- python $R/tools/msda_probe.py --shape $shape --dir bwd --variant 900 --iters 50 > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/lvlprof/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"] or "fillBuffer" in r["Name"]:
            print("$shape", r["Name"].replace("(anonymous namespace)::","").split("(")[0][-60:], r["Calls"], "avg %.1f us" % (float(r["AverageNs"])/1e3))
PY
done
