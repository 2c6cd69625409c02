#!/bin/bash
# round 6: COCO-Full (five levels) step by product build and IO form.  LIBS="prev cur"
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep3.so
for lib in $LIBS; do cp ab/lib_$lib.so semi-detr_amd/csrc/libsemidetr_hip.so
for io in "--io locattn --unmasked" "--io raw --unmasked" "--io raw --masked"; do
timeout 600 python bench.py --recipe full --no-cpu-baseline --no-micro --no-flavours --steps 8 $io > gpurun_out/full_$lib.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("gpurun_out/full_$lib.json").read().strip().splitlines()[-1]); g = d["rooflines_all_msda_groups"]
print("[full $lib $io] step %.3f ms" % d["ms_per_step"], {k.replace("msda_", ""): round(v["avg_launch_us"], 1) for k, v in g.items() if "enc" in k})
PY
done; done
cp /tmp/lib_keep3.so semi-detr_amd/csrc/libsemidetr_hip.so
if [ -n "$TESTS" ]; then timeout -k 5 2400 python -m pytest $TESTS -x -q 2>&1 | tail -4; fi
