#!/bin/bash
# cooperative zero fill inside the merged backward launch (variant 902) against the default (memset + merged launch)
cd $GRAFT_REPO_ROOT
SEMIDETR_TEST_VARIANT=0,902 timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "decoder or five_level_bs2" 2>&1 | tail -1
timeout 600 python tools/stress_parity.py --cases 300 --seed 21 --variant 0 902 2>&1 | tail -2
for rep in 1 2; do
for v in 0 902; do
  timeout 120 python tools/msda_probe.py --shape micro --bs 2 --dir bwd --variant $v --iters 300 2>&1 | tail -1
  timeout 120 python tools/msda_probe.py --shape dec --bs 1 --dir bwd --variant $v --iters 100 2>&1 | tail -1
  timeout 120 python tools/msda_probe.py --shape dec --bs 4 --dir bwd --variant $v --iters 100 2>&1 | tail -1
done
done
timeout 60 python - <<PY
import ctypes, semi_detr_amd as sda
buf = (ctypes.c_ulonglong * 16)()
sda._lib.lib().semidetr_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 0)
print("wait timeouts (must be 0):", buf[15])
PY
