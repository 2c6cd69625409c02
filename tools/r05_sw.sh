#!/bin/bash
# round 5: cell-sorted lane-per-row scatter (msda_sw_d32) -- parity of every encoder-backward test with the product library swapped, then timing
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so
cp ab/lib_${1:-sw}.so semi-detr_amd/csrc/libsemidetr_hip.so
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_msda.py tests/test_gpu_fused.py tests/test_gpu_window_gather.py tests/test_gpu_sweep.py tests/test_gpu_module.py -q 2>&1 | tail -15
cp /tmp/lib_keep.so semi-detr_amd/csrc/libsemidetr_hip.so
LIBS="${1:-sw}" bash tools/r05_ab_kern.sh
LIBS="${1:-sw}" BS=1 bash tools/r05_ab_kern.sh
