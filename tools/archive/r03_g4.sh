#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
export SEMIDETR_EXPERIMENTS=0
SEMIDETR_G4=0 timeout 900 python -m pytest tests/test_gpu_msda.py tests/test_gpu_fullsize.py tests/test_gpu_fused.py -q -m gpu 2>&1 | tail -3
for G in 2 0 1; do
export SEMIDETR_G4=$G
echo "g4 $G: $(python tools/msda_probe.py --shape enc --bs 4 --dir bwd --iters 30 | tail -1)  | $(python tools/msda_probe.py --shape enc --bs 1 --dir bwd --iters 30 | tail -1 | cut -c30-50)"
done
cd /tmp; export TMPDIR=/tmp SEMIDETR_G4=0
rm -rf $R/gpurun_out/g4; rocprofv3 --kernel-trace --stats -d $R/gpurun_out/g4 --output-format csv -- python $R/tools/msda_probe.py --shape enc --bs 4 --dir bwd --iters 20 > /dev/null 2>&1
f=$(find $R/gpurun_out/g4 -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "msda" in r["Name"]: print(r["Name"][28:80], r["Calls"], r["AverageNs"])
PY
