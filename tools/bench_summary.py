#!/usr/bin/env python
"""Short summary of a bench.py JSON line:  python tools/bench_summary.py gpurun_out/x.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: round(d[k], 2) for k in ("value", "ms_per_step")}, "frac", round(d["roofline"]["frac"], 4), d["roofline"]["kernel"])
print({k: round(v, 3) for k, v in d["breakdown_ms_per_step"].items()})
print({k: (round(v["ms_per_step"], 2), round(v["images_per_s"], 1)) for k, v in d.get("flavours", {}).items() if isinstance(v, dict)})
print({k: round(v, 2) for k, v in d.items() if k.startswith("microbench_cold") and isinstance(v, float) and k.endswith("_us")})
