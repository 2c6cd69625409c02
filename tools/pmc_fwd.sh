#!/bin/bash
# usage: tools/pmc_fwd.sh <probe args...> ; collects a few TA/TCP/TCC/SQ counter sets for the msda kernels
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "TA_BUSY_avr" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pmcset_$i
  timeout -k 5 90 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmcset_$i -- python $R/tools/msda_probe.py "$@" > $R/gpurun_out/pmcset_$i.log 2>&1 || tail -3 $R/gpurun_out/pmcset_$i.log
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmcset_*/**/*_counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[1][-30:] if False else r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(f"{k[0][-40:]:42s} {k[1]:40s} {sum(v)/len(v):16.1f}  n={len(v)}")
PY
