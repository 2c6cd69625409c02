#!/bin/bash
# Evidence for `roofline_l1` in the bench line (round 4): TA / TCP / TCC / SQ counters and GRBM_GUI_ACTIVE (effective clock =
# active cycles / kernel time) of BOTH encoder forward kernels at bs 4 (policy patch / window, probe inputs sigma 2 px)
# -> gpurun_out/r06_fwd_enc_TA.txt and .json (bench.py reads the observed clock and the TA busy fraction of the kernel that ran
# from profiles/r06_fwd_enc_TA.json instead of carrying constants).
cd $GRAFT_REPO_ROOT
: > gpurun_out/r06_fwd_enc_TA.txt
for pol in patch window; do
{
echo "# encoder forward, bs 4, policy $pol, probe inputs (sigma 2 px); one rocprofv3 --pmc pass per counter set"
SEMIDETR_EXPERIMENTS=0 bash tools/pmc_fwd.sh --shape enc --bs 4 --dir fwd --variant 0 --policy $pol --iters 5 --io raw --masked 2>&1 | grep -v "^$" | grep "$( [ $pol = patch ] && echo msda_fwd_d32 || echo msda_rw_d32 )" | tail -14
echo "# kernel time of the same launches (rocprofv3 --kernel-trace --stats, no counters)"
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ta_stats
SEMIDETR_EXPERIMENTS=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ta_stats -- python $GRAFT_REPO_ROOT/tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --policy $pol --iters 10 --io raw --masked > /dev/null 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/ta_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "msda_" in r["Name"]: print("KERNEL_AVG %s %.1f us" % (r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("<")[0], float(r["AverageNs"]) / 1e3))
PY
cd $GRAFT_REPO_ROOT
} >> gpurun_out/r06_fwd_enc_TA.txt 2>&1
done
python - <<'PY'
import json, re
txt = open("gpurun_out/r06_fwd_enc_TA.txt").read()
out = {"source": "tools/r06_fwd_ta_evidence.sh: rocprofv3 --pmc passes (tools/pmc_fwd.sh) + a --kernel-trace --stats pass of "
                 "tools/msda_probe.py per forward policy; observed_clock_mhz = GRBM_GUI_ACTIVE / 8 XCDs / kernel time, "
                 "ta_busy_frac = TA_BUSY_avr / (GRBM_GUI_ACTIVE / 8)"}
for kern in ("msda_fwd_d32", "msda_rw_d32"):
    def val(name):
        m = re.search(r"%s[^\n]*?\s%s\s+([0-9.]+)" % (kern, name), txt)
        return float(m.group(1)) if m else None
    avg = re.search(r"KERNEL_AVG %s ([0-9.]+) us" % kern, txt)
    grbm, ta, us = val("GRBM_GUI_ACTIVE"), val("TA_BUSY_avr"), float(avg.group(1)) if avg else None
    out[kern] = {"avg_launch_us": us, "grbm_gui_active": grbm, "ta_busy_avr": ta,
                 "observed_clock_mhz": grbm / 8 / us if grbm and us else None,
                 "ta_busy_frac": ta / (grbm / 8) if grbm and ta else None}
json.dump(out, open("gpurun_out/r06_fwd_enc_TA.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
