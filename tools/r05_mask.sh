#!/bin/bash
# round 5: padding mask inside the region-window forward -- parity, then the masked step flavours and the module-level cost
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r05_mask; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_forward_policy.py tests/test_gpu_fused.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for f in "--io raw" "--io raw --masked" "--io locattn" "--io locattn --masked"; do
  timeout 600 python bench.py --no-cpu-baseline --no-micro --no-flavours --steps 10 $f > "$O/bench_$(echo $f | tr -d ' -').json" 2> $O/err.log || tail -5 $O/err.log
done
python - <<'PY' 2>&1 | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_mask/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    g = d.get("rooflines_all_msda_groups", {})
    print(f.split("/")[-1], "ms/step %.3f" % d["ms_per_step"], "img/s %.1f" % d["value"])
    for k, v in sorted(g.items()):
        print("   ", k, "%.1f us" % v["avg_launch_us"], v.get("kernels", ""))
PY
timeout 600 python - <<'PY' 2>&1 | tee $O/module.txt
import torch, bench, json
r = bench.module_bench(torch.device("cuda", 0))
print(json.dumps({k: v for k, v in r.items() if k != "what"}, indent=1))
PY
