#!/bin/bash
# A/B of two builds of the product library on the BASELINE micro-benchmark (graph replays, warm and cold) + secondary shapes
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for which in old new; do
cp ab/lib$which.so semi-detr_amd/csrc/libsemidetr_hip.so
timeout 600 python - <<PY
import torch, semi_detr_amd, bench
r = bench.microbench(torch.device("cuda:0"))
print("[$which $rep] warm", {k: round(r[k], 2) for k in ("fwd_us", "bwd_us", "fwd_bwd_us")},
      "cold", {k: (round(v["us"], 2), round(v["frac_hbm_measured"], 3)) for k, v in r["cold"].items() if isinstance(v, dict)},
      "secondary bwd", {k: round(v["bwd_us"], 1) for k, v in r["secondary_shapes"].items()})
PY
done; done
cp ab/libnew.so semi-detr_amd/csrc/libsemidetr_hip.so
