#!/bin/bash
# SQ counters of the encoder backward's kernels for builds of the product library (ab/lib_<name>.so): what is different about a build?
R=$GRAFT_REPO_ROOT
cd $R; cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so
cd /tmp && export TMPDIR=/tmp
for which in $LIBS; do
cp $R/ab/lib_$which.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcs_${which}_$i
  SEMIDETR_EXPERIMENTS=0 timeout -k 5 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcs_${which}_$i -- python $R/tools/msda_probe.py --shape enc --bs 4 --dir bwd --iters 4 --cold 4 --variant 0 > $R/gpurun_out/pmcs_${which}_$i.log 2>&1 || tail -3 $R/gpurun_out/pmcs_${which}_$i.log
done
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(list)
for f in sorted(glob.glob("$R/gpurun_out/pmcs_${which}_*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "scatter" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("[$which]", {k: round(sum(v)/len(v)/1e6, 2) for k, v in agg.items()}, "(millions per launch)")
PY
done
cp /tmp/lib_keep.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
