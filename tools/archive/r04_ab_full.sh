#!/bin/bash
# A/B of several builds of the product library inside the COCO-Full bench step (--recipe full: five levels)
cd $GRAFT_REPO_ROOT
cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so
for rep in 1 2; do for which in $LIBS; do
cp ab/lib_$which.so semi-detr_amd/csrc/libsemidetr_hip.so
timeout 600 python bench.py --steps 8 --warmup 3 --no-micro --no-flavours --no-cpu-baseline --recipe full > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
b = d["breakdown_ms_per_step"]
print("[$which full $rep]", round(d["value"], 1), round(d["ms_per_step"], 3), {k: round(v, 3) for k, v in b.items() if "enc" in k})
PY
done; done
cp /tmp/lib_keep.so semi-detr_amd/csrc/libsemidetr_hip.so
