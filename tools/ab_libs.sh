#!/bin/bash
# A/B of two builds of libsemidetr_hip.so on ONE box: ab/libold.so vs ab/libnew.so, per-kernel averages under rocprofv3,
# interleaved twice.  usage: ab_libs.sh [probe args ...]   (default: encoder bs 4 backward)
R=$GRAFT_REPO_ROOT
ARGS=${@:---shape enc --bs 4 --dir bwd --variant 0 --iters 12}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for which in old new; do
  cp $R/ab/lib$which.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
  rm -rf $R/gpurun_out/ab_$which
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_$which -- python $R/tools/msda_probe.py $ARGS > $R/gpurun_out/ab_$which.log 2>&1
  tail -1 $R/gpurun_out/ab_$which.log | sed "s/^/[$which $rep] /"
  python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/ab_$which/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"] or "fill" in r["Name"]:
            print("   [$which $rep] %-60s avg %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-58:], float(r["AverageNs"]) / 1e3))
PY
done
done
cp $R/ab/libnew.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
