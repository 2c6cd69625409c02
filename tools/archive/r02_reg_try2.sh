#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "encoder_self_attention" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 65 69; do
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf $R/gpurun_out/regev_${v}_$c
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/regev_${v}_$c -- python $R/tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 3 > /dev/null 2>&1
  done
done
rm -rf $R/gpurun_out/regev_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/regev_stats -- python $R/tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 69 --iters 10 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
for v in (65, 69):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for f in glob.glob("$R/gpurun_out/regev_%d_%s/**/*_counter_collection.csv" % (v, c), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if "scatter" in k and r["Counter_Name"] == c:
                    agg[k.replace("(anonymous namespace)::", "").split("(")[0][-50:]].append(float(r["Counter_Value"]))
        for k, vals in agg.items():
            print("variant %d %-11s %-52s %10.1f KiB/launch" % (v, c, k, sum(vals) / len(vals)))
for f in glob.glob("$R/gpurun_out/regev_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"]:
            print("variant 69: %-60s avg %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-58:], float(r["AverageNs"]) / 1e3))
PY
