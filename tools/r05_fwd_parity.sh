#!/bin/bash
# forward-kernel parity of the product library given as $1 (ab/lib_<name>.so), then restore
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so; cp ab/lib_$1.so semi-detr_amd/csrc/libsemidetr_hip.so
timeout 1200 python -m pytest tests/test_gpu_forward_policy.py tests/test_gpu_fullsize.py tests/test_gpu_fused.py tests/test_gpu_module.py tests/test_gpu_msda.py -x -q -m gpu 2>&1 | tail -5
cp /tmp/lib_keep.so semi-detr_amd/csrc/libsemidetr_hip.so
