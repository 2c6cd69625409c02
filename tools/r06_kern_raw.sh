#!/bin/bash
# round 6: per-kernel rocprofv3 averages on the HEADLINE's inputs (fused prologue + mask, bench.Workload) for product builds ab/lib_<name>.so.
#   LIBS="a b" DIR=bwd|fwd BS=4 POLICY=window bash tools/r06_kern_raw.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for which in $LIBS; do
cp $R/ab/lib_$which.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
rm -rf $R/gpurun_out/abk_$which
SEMIDETR_EXPERIMENTS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abk_$which -- python $R/tools/msda_probe.py --shape ${SHAPE:-enc} --bs ${BS:-4} --dir ${DIR:-bwd} --iters 12 --cold 6 --policy ${POLICY:-window} --io raw ${MASKED---masked} > $R/gpurun_out/abk_$which.log 2>&1
grep "us  alg" $R/gpurun_out/abk_$which.log | sed "s/^/[$which] /"
python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/abk_$which/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"] and "mask_extents" not in r["Name"]:
            print("   [$which] %-64s calls %s avg %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-62:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $R/gpurun_out/abk_$which
done
cp /tmp/lib_keep.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
