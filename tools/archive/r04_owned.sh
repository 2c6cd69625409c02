#!/bin/bash
# round 4: owned-rows backward (no fill, no atomics) for query sets that fit one workgroup: parity + micro-benchmark timing
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_msda.py tests/test_gpu_fused.py tests/test_gpu_module.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -4
for rep in 1 2; do
SEMIDETR_EXPERIMENTS=0 timeout 120 python tools/msda_probe.py --shape micro --bs 2 --dir bwd --iters 200 --cold 8 --print-kernels 2>&1 | tail -2 | tr '\n' ' '; echo
SEMIDETR_EXPERIMENTS=0 timeout 120 python tools/msda_probe.py --shape dec --bs 4 --lq 300 --dir bwd --iters 100 --cold 4 --print-kernels 2>&1 | tail -2 | tr '\n' ' '; echo
SEMIDETR_EXPERIMENTS=0 timeout 120 python tools/msda_probe.py --shape dec --bs 4 --lq 1100 --dir bwd --iters 100 --cold 4 --print-kernels 2>&1 | tail -2 | tr '\n' ' '; echo
done
timeout 600 python - <<'PY'
import torch, bench
r = bench.microbench(torch.device("cuda:0"))
print({k: r[k] for k in ("fwd_us", "bwd_us", "fwd_bwd_us", "frac_hbm_measured")})
print("cold", {k: (round(v["us"], 2), round(v["frac_hbm_measured"], 3)) for k, v in r["cold"].items() if isinstance(v, dict)})
print("secondary", {k: (round(v["fwd_us"], 1), round(v["bwd_us"], 1)) for k, v in r["secondary_shapes"].items()})
PY
