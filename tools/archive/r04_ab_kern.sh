#!/bin/bash
# per-kernel rocprofv3 averages of the encoder backward for several builds of the PRODUCT library (ab/lib_<name>.so)
R=$GRAFT_REPO_ROOT
cd $R; cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so
cd /tmp && export TMPDIR=/tmp
for which in $LIBS; do
cp $R/ab/lib_$which.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
rm -rf $R/gpurun_out/abk_$which
SEMIDETR_EXPERIMENTS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abk_$which -- python $R/tools/msda_probe.py --shape enc --bs ${BS:-4} --dir bwd --iters 12 --cold 6 --variant 0 > $R/gpurun_out/abk_$which.log 2>&1
grep "us  alg" $R/gpurun_out/abk_$which.log | sed "s/^/[$which] /"
python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/abk_$which/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"] or "fill" in r["Name"]:
            print("   [$which] %-64s calls %s avg %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-62:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cp /tmp/lib_keep.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
