#!/bin/bash
# round 6: window gather with the round's grad_out rows staged in LDS -- parity, per-kernel averages (rocprofv3) base vs new, then the step
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_window_gather.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4
LIBS="${LIBS:-base gwl}" POLICY=window bash tools/r05_ab_kern.sh 2>&1 | grep "gw_d32\|scatter_d32_reg\|us  alg"
EXTRA="--no-flavours" LIBS="${LIBS:-base gwl}" REPS=2 bash tools/r05_ab_step.sh 2>&1 | grep "^\["
for lib in ${LIBS:-base gwl}; do
cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep2.so; cp ab/lib_$lib.so semi-detr_amd/csrc/libsemidetr_hip.so
timeout 600 python bench.py --recipe full --no-cpu-baseline --no-micro --no-flavours --steps 10 > gpurun_out/gwl_full_$lib.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("gpurun_out/gwl_full_$lib.json").read().strip().splitlines()[-1]); b = d["breakdown_ms_per_step"]
print("[full $lib] step %.3f ms" % d["ms_per_step"], {k: round(v, 3) for k, v in b.items() if "enc" in k})
PY
cp /tmp/lib_keep2.so semi-detr_amd/csrc/libsemidetr_hip.so
done
