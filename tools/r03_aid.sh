#!/bin/bash
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp SEMIDETR_EXPERIMENTS=0
for G in 2 0 3 4; do
export SEMIDETR_G4=$G
rm -rf $R/gpurun_out/g4; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/g4 --output-format csv -- python $R/tools/msda_probe.py --shape enc --bs 4 --dir bwd --iters 20 > /dev/null 2>&1
f=$(find $R/gpurun_out/g4 -name "*kernel_stats.csv" | head -1); echo "g4 $G"; python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "gather" in r["Name"]: print("   ", r["Name"][28:100], r["Calls"], r["AverageNs"])
PY
done
