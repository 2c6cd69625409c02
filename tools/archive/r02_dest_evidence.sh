#!/bin/bash
# evidence for the destination-owned grad_value kernel (variant 70) vs the windowed one (65): time + HBM-side traffic
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r02_dest_owned_scatter.txt
echo "# encoder backward bs 4 (N=4, Lq=S=22223), tools/r02_dest_evidence.sh" > $OUT
for v in 65 70 71 72; do python $R/tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 20 2>&1 | tail -1 >> $OUT; done
for v in 65 70; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/destev_${v}_$c
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/destev_${v}_$c -- python $R/tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant $v --iters 3 > /dev/null 2>&1
  done
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/destev_stats -- python $R/tools/msda_probe.py --shape enc --bs 4 --dir bwd --variant 70 --iters 10 > /dev/null 2>&1
python - <<PY >> $OUT
import csv, glob, collections
for v in (65, 70):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = collections.defaultdict(list)
        for f in glob.glob("$R/gpurun_out/destev_%d_%s/**/*_counter_collection.csv" % (v, c), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if ("msda_" in k or "fillBuffer" in k) and r["Counter_Name"] == c:
                    agg[k.replace("(anonymous namespace)::", "").split("(")[0][-60:]].append(float(r["Counter_Value"]))
        for k, vals in agg.items():
            print("variant %d %-11s %-62s %10.1f KiB/launch (n=%d)" % (v, c, k, sum(vals) / len(vals), len(vals)))
for f in glob.glob("$R/gpurun_out/destev_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"] or "fillBuffer" in r["Name"]:
            print("kernel-trace variant 70: %-70s calls %s avg %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-68:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cat $OUT
