#!/bin/bash
# GPU suite + the bench in its four flavours (default, --io raw, --recipe full, --recipe full --io raw) + traffic counters
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/r03_suite.log 2>&1
tail -6 $O/r03_suite.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/r03_bench_default.json 2> $O/r03_bench_default.err
timeout 600 python bench.py --steps 10 --warmup 3 --io raw --no-cpu-baseline > $O/r03_bench_raw.json 2> $O/r03_bench_raw.err
timeout 600 python bench.py --steps 10 --warmup 3 --recipe full --no-cpu-baseline > $O/r03_bench_full.json 2> $O/r03_bench_full.err
timeout 600 python bench.py --steps 10 --warmup 3 --recipe full --io raw --no-cpu-baseline > $O/r03_bench_full_raw.json 2> $O/r03_bench_full_raw.err
python - <<'PY'
import json
for n in ("default", "raw", "full", "full_raw"):
    try:
        d = json.loads(open(f"gpurun_out/r03_bench_{n}.json").read().strip().splitlines()[-1])
        print(n, {k: round(d[k], 2) for k in ("value", "ms_per_step")}, "frac", round(d["roofline"]["frac"], 4), d["roofline"]["kernel"])
        print("   ", {k: round(v, 2) for k, v in d["breakdown_ms_per_step"].items()})
        if "microbench" in d:
            m = d["microbench"]
            print("    micro warm", {k: (round(v["us"], 2), round(v["frac_hbm_measured"], 3)) for k, v in m["warm"].items() if isinstance(v, dict)})
            print("    micro cold", {k: (round(v["us"], 2), round(v["frac_hbm_measured"], 3)) for k, v in m["cold"].items() if isinstance(v, dict)})
    except Exception as e:
        print(n, "FAILED", e, open(f"gpurun_out/r03_bench_{n}.err").read()[-800:])
PY
