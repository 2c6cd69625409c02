#!/bin/bash
# which encoder-forward kernel the adaptive policy launched, launch by launch, over a bench run (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/at
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/at -- python $R/bench.py --steps 10 --warmup 3 --no-micro --no-flavours --no-cpu-baseline > $R/gpurun_out/at.json 2> $R/gpurun_out/at.err
python - <<PY
import csv, glob, json
for f in glob.glob("$R/gpurun_out/at/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f))]
    seq = []
    for r in rows:
        n = r["Kernel_Name"]
        g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
        if "msda_rw_d32" in n: seq.append(("W", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        elif "msda_fwd_d32" in n and "408" in n and g > 256 * 20000: seq.append(("P", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    print("bs4 encoder forward launches:", len(seq))
    print("".join(k for k, _ in seq))
    for kind in "PW":
        d = [t for k, t in seq if k == kind]
        if d: print(kind, len(d), "mean %.1f us" % (sum(d) / len(d)))
d = json.loads(open("$R/gpurun_out/at.json").read().strip().splitlines()[-1])
print(d["value"], d["forward_policy"])
PY
