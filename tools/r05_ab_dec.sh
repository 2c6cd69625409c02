#!/bin/bash
# decoder backward (Lq 1100) per build: rocprofv3 kernel averages
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so
cd /tmp && export TMPDIR=/tmp
for which in $LIBS; do
cp $R/ab/lib_$which.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
rm -rf $R/gpurun_out/abd_$which
SEMIDETR_EXPERIMENTS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abd_$which -- python $R/tools/msda_probe.py --shape dec --lq 1100 --bs ${BS:-4} --dir bwd --iters 20 --cold 6 --variant 0 > $R/gpurun_out/abd_$which.log 2>&1
grep "us  alg" $R/gpurun_out/abd_$which.log | sed "s/^/[$which] /"
python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/abd_$which/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"]:
            print("   [$which] %-50s calls %s avg %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-48:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/lib_keep.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
