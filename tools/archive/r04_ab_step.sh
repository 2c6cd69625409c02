#!/bin/bash
# A/B of two builds of the product library inside the bench step: ab/libold.so vs ab/libnew.so, interleaved twice on one box
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for which in old new; do
cp ab/lib$which.so semi-detr_amd/csrc/libsemidetr_hip.so
timeout 600 python bench.py --steps 20 --warmup 5 --no-micro --no-flavours --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
b = d["breakdown_ms_per_step"]
print("[$which $rep]", round(d["value"], 1), round(d["ms_per_step"], 3), {k.replace("msda_", ""): round(v, 3) for k, v in b.items() if "enc" in k or "dec_bs4" in k})
PY
done; done
cp ab/libold.so semi-detr_amd/csrc/libsemidetr_hip.so
