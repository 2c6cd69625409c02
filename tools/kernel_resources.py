#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of a .hip source (hipcc -Rpass-analysis=kernel-resource-usage), one line per
kernel.   python tools/kernel_resources.py [regex on the demangled name] [--src msda.hip] [--extra "-DX=1"]"""
import argparse
import os
import re
import subprocess

ap = argparse.ArgumentParser()
ap.add_argument("pattern", nargs="?", default="")
ap.add_argument("--src", default="msda.hip")
ap.add_argument("--extra", default="")
a = ap.parse_args()
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "semi-detr_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-I../../include",
       "-Wno-unused-function", "-Wno-pass-failed", "-Wno-unused-variable", "-Rpass-analysis=kernel-resource-usage", "-c", a.src,
       "-o", "/tmp/_kr.o"] + a.extra.split()
txt = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
rows, cur = [], None
for line in txt.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True,
                       text=True).stdout.split("\n")
seen = set()
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if (a.pattern and not re.search(a.pattern, n)) or n in seen:
        continue
    seen.add(n)
    g = lambda k: r.get(k, "?")
    print(f"{n[:100]:100s} vgpr {g('VGPRs'):>4} agpr {g('AGPRs'):>3} sgpr {g('SGPRs'):>4} spill v{g('VGPRs Spill')}/s{g('SGPRs Spill')} "
          f"scratch {g('ScratchSize [bytes/lane]'):>4} occ {g('Occupancy [waves/SIMD]')} lds {g('LDS Size [bytes/block]')}")
print(f"# {len(seen)} kernels")
