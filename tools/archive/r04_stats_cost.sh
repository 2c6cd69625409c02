#!/bin/bash
# round 4: what the far-sample counting costs inside the kernels: forced policy (no counting) vs adaptive after settling, kernel
# durations from rocprofv3 (no host effects), sigma 2 px (window) and 4 px (patch), rotated inputs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "2.0 window" "2.0 adaptive" "4.0 patch" "4.0 adaptive"; do
set -- $cfg
rm -rf $R/gpurun_out/sc
SEMIDETR_EXPERIMENTS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sc -- python $R/tools/msda_probe.py --shape enc --bs 4 --dir fwd --iters 48 --sigma $1 --cold 6 --policy $2 --settle 6 > $R/gpurun_out/sc.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/sc/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "msda_" in r["Kernel_Name"]]
    last = rows[-40:]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last]
    names = sorted(set(r["Kernel_Name"].split("(")[0][-40:] for r in last))
    d.sort()
    print("[sigma $1 $2 rep $rep] last 40 launches: median %.1f us  mean %.1f  p90 %.1f  %s" % (d[len(d)//2], sum(d)/len(d), d[int(len(d)*0.9)], names))
PY
done; done
