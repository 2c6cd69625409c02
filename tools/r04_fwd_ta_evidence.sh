#!/bin/bash
# Evidence for `roofline_l1` in the bench line (round 4): TA / TCP / TCC / SQ counters and GRBM_GUI_ACTIVE (effective clock =
# active cycles / kernel time) of the encoder forward at bs 4 -> gpurun_out/r04_fwd_enc_TA.txt and .json (bench.py reads the
# observed clock and the TA busy fraction from profiles/r04_fwd_enc_TA.json instead of carrying constants).
cd $GRAFT_REPO_ROOT
{
echo "# encoder forward, bs 4, probe inputs (sigma 2 px); one rocprofv3 --pmc pass per counter set"
bash tools/pmc_fwd.sh --shape enc --bs 4 --dir fwd --variant 0 --iters 5 2>&1 | grep -v "^$" | tail -12
echo "# kernel time of the same launches (rocprofv3 --kernel-trace --stats, no counters)"
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ta_stats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ta_stats -- python $GRAFT_REPO_ROOT/tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --iters 10 > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/ta_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"]: print("%-60s avg %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").split("(")[0][-58:], float(r["AverageNs"]) / 1e3))
PY
cd $GRAFT_REPO_ROOT
} > gpurun_out/r04_fwd_enc_TA.txt 2>&1
python - <<'PY'
import json, re
txt = open("gpurun_out/r04_fwd_enc_TA.txt").read()
def val(name):
    m = re.search(r"msda_fwd_d32.*?\s%s\s+([0-9.]+)" % name, txt)
    return float(m.group(1)) if m else None
avg = re.search(r"msda_fwd_d32<[^>]*>\s+avg ([0-9.]+) us", txt)
grbm, ta = val("GRBM_GUI_ACTIVE"), val("TA_BUSY_avr")
us = float(avg.group(1)) if avg else None
out = {"kernel": "msda_fwd_d32 (encoder forward, bs 4, probe inputs sigma 2 px)", "avg_launch_us": us, "grbm_gui_active": grbm,
       "ta_busy_avr": ta, "observed_clock_mhz": grbm / 8 / us if grbm and us else None,
       "ta_busy_frac": ta / (grbm / 8) if grbm and ta else None,
       "source": "tools/r04_fwd_ta_evidence.sh: rocprofv3 --pmc passes (tools/pmc_fwd.sh) + a --kernel-trace --stats pass of tools/msda_probe.py"}
json.dump(out, open("gpurun_out/r04_fwd_enc_TA.json", "w"), indent=1)
print(out)
PY
cat gpurun_out/r04_fwd_enc_TA.txt
