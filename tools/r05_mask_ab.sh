#!/bin/bash
# masked vs unmasked encoder forward for several builds of the product library (ab/lib_<name>.so), same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so
for rep in 1 2; do
for which in $LIBS; do
  cp ab/lib_$which.so semi-detr_amd/csrc/libsemidetr_hip.so
  echo "[$which $rep] $(timeout 300 python tools/r05_mask_probe.py ${POLICY:-window} 2>&1 | tail -1)"
done
done
cp /tmp/lib_keep.so semi-detr_amd/csrc/libsemidetr_hip.so
