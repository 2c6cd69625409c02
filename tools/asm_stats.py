#!/usr/bin/env python3
"""Static instruction mix of every kernel in a device assembly file (hipcc ... --offload-device-only -S).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude --offload-device-only -S \
          semi-detr_amd/csrc/msda.hip -o /tmp/msda.s && python tools/asm_stats.py /tmp/msda.s [name-filter]

Counts are static (loops count once); they answer "did the compiler pack the fp32 FMAs" and "how many LDS / vector-memory
instructions does the unrolled body hold", not run time.
"""
import collections
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    cur, mcur, cnt, meta = None, None, collections.defaultdict(collections.Counter), collections.defaultdict(dict)
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^\s*\.name:\s*(_Z\w+)", line)      # metadata: .name precedes the entry's register counts
        if m:
            mcur = m.group(1)
            continue
        m = re.match(r"^\s*\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size):\s*(\d+)", line)
        if m and mcur:
            meta[mcur][m.group(1)] = int(m.group(2))
            continue
        if cur and line.startswith("\t"):
            t = line.split()
            if not t or t[0][0] in ".;":
                continue
            op = t[0]
            c = cnt[cur]
            if op.startswith("v_pk_"):
                c[op] += 1
            elif op.startswith(("v_fma_f32", "v_fmac_f32")):
                c["fma"] += 1
            elif op.startswith("v_"):
                c["valu_other"] += 1
            elif op.startswith("ds_"):
                c["ds"] += 1
            elif op.startswith(("buffer_", "global_", "scratch_")):
                c["vmem"] += 1
            elif op.startswith("s_"):
                c["salu"] += 1
    names = demangle(list(cnt))
    for k, c in cnt.items():
        n = names.get(k, k)
        if flt and flt not in n:
            continue
        print(n[:110])
        print("   ", dict(c), meta.get(k, {}))


if __name__ == "__main__":
    main()
