#!/bin/bash
# in-step A/B of product builds (ab/lib_<name>.so): bench.py's default step (no micro-benchmark, no CPU baseline), REPS interleaved passes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so
for rep in $(seq 1 ${REPS:-2}); do
for which in $LIBS; do
cp ab/lib_$which.so semi-detr_amd/csrc/libsemidetr_hip.so
timeout 600 python bench.py --no-cpu-baseline --no-micro $EXTRA > gpurun_out/abs_$which.json 2> gpurun_out/abs_$which.err
python - <<PY
import json
d = json.loads(open("gpurun_out/abs_$which.json").read().strip().splitlines()[-1])
b = d["breakdown_ms_per_step"]
print("[$which $rep] step %.3f ms | enc fwd bs4 %.3f bs1 %.3f | enc bwd bs4 %.3f bs1 %.3f |" % (d["ms_per_step"], b["msda_fwd_enc_bs4_Lq22223"], b["msda_fwd_enc_bs1_Lq22223"], b["msda_bwd_enc_bs4_Lq22223"], b["msda_bwd_enc_bs1_Lq22223"]),
      {k: round(v["ms_per_step"], 2) for k, v in d.get("flavours", {}).items() if isinstance(v, dict)})
PY
done
done
cp /tmp/lib_keep.so semi-detr_amd/csrc/libsemidetr_hip.so
