#!/bin/bash
# SQ-level counters (one small set per pass, each pass under `timeout`) for the msda kernels of a probe run.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/sqset_$i -- python $R/tools/msda_probe.py "$@" > $R/gpurun_out/sqset_$i.log 2>&1 || tail -2 $R/gpurun_out/sqset_$i.log
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/sqset_*/**/*_counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(f"{k[0][-36:]:38s} {k[1]:28s} {sum(v)/len(v):16.1f}  n={len(v)}")
PY
