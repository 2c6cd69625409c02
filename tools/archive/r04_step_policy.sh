#!/bin/bash
# round 4: the bench step under the three forward policies (same box, interleaved twice)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for pol in patch window adaptive; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-micro --no-flavours --no-cpu-baseline --forward-policy $pol > gpurun_out/sp.json 2> gpurun_out/sp.err
python - <<PY
import json
d = json.loads(open("gpurun_out/sp.json").read().strip().splitlines()[-1])
b = d["breakdown_ms_per_step"]
print("[$pol $rep]", round(d["value"], 1), round(d["ms_per_step"], 3), "enc fwd bs4 us", round(b["msda_fwd_enc_bs4_Lq22223"] / 24 * 1e3, 1), d["forward_policy"], d["roofline"]["kernel_symbols"])
PY
done; done
