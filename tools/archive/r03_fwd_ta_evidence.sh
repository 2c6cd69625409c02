#!/bin/bash
# Evidence for `roofline_l1` in the bench line: TA / TCP / TCC / SQ counters and GRBM_GUI_ACTIVE (effective clock = active
# cycles / kernel time) of the default encoder forward at bs 4, then the out-of-range probe (loads that never touch the cache).
cd $GRAFT_REPO_ROOT
{
echo "# default encoder forward msda_fwd_d32<1, 4, 408>, bs 4, probe inputs (sigma 2 px); one rocprofv3 --pmc pass per counter set"
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
echo "# tools/r02_oob_probe.py: the same forward with a share of the samples pushed outside the map (their loads return zeros without touching the cache)"
timeout 300 python tools/r02_oob_probe.py 2>&1 | grep fraction
} > gpurun_out/r03_fwd_enc_TA.txt 2>&1
cat gpurun_out/r03_fwd_enc_TA.txt
{
echo "# --- the two encoder backward kernels, same counters"
bash tools/pmc_fwd.sh --shape enc --bs 4 --dir bwd --variant 0 --iters 5 2>&1 | grep -v "^$" | tail -22
} >> gpurun_out/r03_fwd_enc_TA.txt 2>&1
tail -24 gpurun_out/r03_fwd_enc_TA.txt
