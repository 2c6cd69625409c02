#!/bin/bash
# round 6: the mask summary kernel (1024 threads, 16 bytes per thread) -- parity, its time under rocprofv3, the step with fresh masks per batch
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_forward_policy.py tests/test_gpu_fused.py -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/mext
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mext -- python $R/bench.py --steps 10 --warmup 3 --no-micro --no-flavours --no-cpu-baseline > $R/gpurun_out/mext.json 2>/dev/null
cd $R; python tools/summarize_prof.py stats gpurun_out/mext | grep -i "mask_extents\|msda_rw_d32\|copy\|elementwise" | head; rm -rf gpurun_out/mext
python - <<'PY'
import json
d = json.loads(open("gpurun_out/mext.json").read().strip().splitlines()[-1]); print(round(d["value"], 1), round(d["ms_per_step"], 3))
PY
