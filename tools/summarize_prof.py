#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small summaries kept under profiles/.
    python tools/summarize_prof.py stats  <dir with *_kernel_stats.csv>  > profiles/rNN_<name>_kernel_stats.txt
    python tools/summarize_prof.py pmc    <dir with *_counter_collection.csv> <COUNTER> > profiles/rNN_<name>_pmc.txt
    python tools/summarize_prof.py bygrid <dir with *_kernel_trace.csv>  > profiles/rNN_<name>_kernel_by_grid.txt
"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)[:70]


def stats(d):
    f = max(glob.glob(d + "/**/*_kernel_stats.csv", recursive=True), key=os.path.getmtime)     # newest run
    rows = list(csv.DictReader(open(f)))
    print(f"# rocprofv3 --kernel-trace --stats ; source {f.split('gpurun_out/')[-1]}")
    print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'pct':>6s}")
    for r in rows[:60]:
        print(f"{short(r['Name']):70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.1f} "
              f"{float(r['TotalDurationNs'])/1e6:10.2f} {float(r['Percentage']):6.2f}")


def pmc(d, counter):
    f = max(glob.glob(d + "/**/*_counter_collection.csv", recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    print(f"# rocprofv3 --pmc {counter} ; per-dispatch average (counter unit: KiB for FETCH_SIZE/WRITE_SIZE)")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) / len(v) > 1000:
            print(f"{k:70s} dispatches {len(v):4d}  avg {sum(v)/len(v):14.1f}  (= {sum(v)/len(v)*1024/1e6:9.1f} MB)")


def bygrid(d):
    """Per (kernel, grid size) averages from the kernel trace: separates the batch sizes one symbol is launched with,
    so the averages can be held against bench.py's event-timed groups."""
    f = max(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if n.startswith(("msda_", "nms_", "o2m_", "tal_", "lsap", "match_cost", "ema_", "pseudo_label", "transform_bboxes",
                         "build_targets")):
            agg[(n, int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]))].append(
                (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"# rocprofv3 --kernel-trace ; per (kernel, workgroups) ; source {f.split('gpurun_out/')[-1]}")
    print(f"{'kernel':62s} {'workgroups':>12s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s}")
    for (n, gx, gy), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if sum(v) > 200:
            print(f"{n:62s} {str(gx) + 'x' + str(gy):>12s} {len(v):6d} {sum(v)/len(v):9.1f} {min(v):9.1f} {max(v):9.1f}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "bygrid": bygrid}[sys.argv[1]](*sys.argv[2:])
