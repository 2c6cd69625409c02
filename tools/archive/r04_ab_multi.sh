#!/bin/bash
# A/B/C of several builds of the product library inside the bench step: LIBS="old t20 t30" (ab/lib<name>.so, ab/lib_<name>.so)
cd $GRAFT_REPO_ROOT
cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so
for rep in 1 2; do for which in $LIBS; do
f=ab/lib$which.so; [ -f $f ] || f=ab/lib_$which.so
cp $f semi-detr_amd/csrc/libsemidetr_hip.so
for io in ${IOS:-locattn raw}; do
timeout 600 python bench.py --steps 12 --warmup 4 --no-micro --no-flavours --no-cpu-baseline --io $io > gpurun_out/ab.json 2> gpurun_out/ab.err
python - <<PY
import json
d = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
b = d["breakdown_ms_per_step"]
print("[$which $io $rep]", round(d["value"], 1), round(d["ms_per_step"], 3), "enc fwd bs4 us", round(b["msda_fwd_enc_bs4_Lq22223"] / 24 * 1e3, 1), "enc bwd bs4 / bs1 us", round(b["msda_bwd_enc_bs4_Lq22223"] / 6 * 1e3, 1), round(b["msda_bwd_enc_bs1_Lq22223"] / 6 * 1e3, 1))
PY
done; done; done
cp /tmp/lib_keep.so semi-detr_amd/csrc/libsemidetr_hip.so
