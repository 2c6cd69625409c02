#!/bin/bash
# per-kernel averages under rocprofv3 for a list of backward variants, interleaved twice on ONE box.
# usage: ab_variants.sh "<variants>" [probe args]      e.g. ab_variants.sh "0 6962" --shape enc --bs 4 --dir bwd
R=$GRAFT_REPO_ROOT
VARS=$1; shift
ARGS=${@:---shape enc --bs 4 --dir bwd --iters 12}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for v in $VARS; do
  rm -rf $R/gpurun_out/abv_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abv_$v -- python $R/tools/msda_probe.py $ARGS --variant $v > $R/gpurun_out/abv_$v.log 2>&1
  grep "us  alg" $R/gpurun_out/abv_$v.log | sed "s/^/[$v $rep] /"
  python - <<PY
import csv, glob
for f in glob.glob("$R/gpurun_out/abv_$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"] or "fill" in r["Name"]:
            print("   [$v $rep] %-64s avg %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-62:], float(r["AverageNs"]) / 1e3))
PY
done
done
