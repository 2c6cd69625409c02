#!/bin/bash
# round 6: encoder forward / backward by sample pattern -- Gaussian spreads and the reference's initial offset star -- under the
# three forward policies; which kernels an untrained and a sigma-grown model reach
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out; rm -f gpurun_out/r06_spread.txt
for lv in 4 5; do
python - $lv <<'PY' | tee -a gpurun_out/r06_spread.txt
import json, torch, bench, semi_detr_amd
import sys
five = len(sys.argv) > 1 and sys.argv[1] == "5"
r = bench.forward_policy_bench(torch.device("cuda:0"), levels=(bench.RECIPES["full"]["levels"] if five else None), sigmas=(1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 8.0))
json.dump(r, open("gpurun_out/r06_spread_%s.json" % ("five" if five else "four"), "w"), indent=1)
print("# %d levels" % (5 if five else 4))
print("%-12s %8s | fwd us: %7s %7s %8s (%s) | bwd us: %7s %7s %8s (%s)" % ("pattern", "far", "patch", "window", "adaptive", "kernel", "patch", "window", "adaptive", "gather"))
for k, v in r.items():
    if not isinstance(v, dict): continue
    print("%-12s %8.3f | %14.1f %7.1f %8.1f (%s) | %14.1f %7.1f %8.1f (%s)" % (k, v["far_fraction"], v["patch_us"], v["window_us"], v["adaptive_us"],
          v["adaptive_kernel"].split("<")[0], v["patch_bwd_us"], v["window_bwd_us"], v["adaptive_bwd_us"], v["adaptive_bwd_kernels"].split("+")[0]))
PY
done
