#!/usr/bin/env python
"""Condense rocprofv3 output directories into the small summaries kept under profiles/.
    python tools/summarize_prof.py stats  <dir with *_kernel_stats.csv>  > profiles/rNN_<name>_kernel_stats.txt
    python tools/summarize_prof.py pmc    <dir with *_counter_collection.csv> <COUNTER> > profiles/rNN_<name>_pmc.txt
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


if __name__ == "__main__":
    (stats if sys.argv[1] == "stats" else pmc)(*sys.argv[2:])
