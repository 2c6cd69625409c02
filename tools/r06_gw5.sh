#!/bin/bash
# round 6: five-level lane-per-sample window gather -- parity, then the COCO-Full step by gather (forward policy patch = patch gather, window = window gather)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06_gw5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_window_gather.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -12 $O/pytest.log
for pol in patch window; do for io in locattn raw; do for mk in --unmasked --masked; do
  [ $io = locattn ] && [ $mk = --masked ] && continue
  timeout 600 python bench.py --recipe full --no-cpu-baseline --no-micro --no-flavours --steps 10 --io $io $mk --forward-policy $pol > $O/bench_${io}${mk}_$pol.json 2> $O/err.log || tail -5 $O/err.log
done; done; done
python - <<'PY' 2>&1 | tee $O/summary.txt
import json, glob
for f in sorted(glob.glob("gpurun_out/r06_gw5/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    g = d.get("rooflines_all_msda_groups", {})
    print(f.split("/")[-1], "ms/step %.3f" % d["ms_per_step"], "img/s %.1f" % d["value"])
    for k, v in sorted(g.items()):
        if "enc" in k:
            print("   ", k, "%.1f us" % v["avg_launch_us"], v.get("kernels", ""))
PY
