#!/bin/bash
# build one variant of the PRODUCT library into ab/lib_<name>.so:  tools/ab_build.sh <name> [-DX=1 ...]
cd /root/repo/semi-detr_amd/csrc
name=$1; shift
mkdir -p /tmp/abb_$name ../../ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I../../include -Wall -Wno-unused-function -Wno-pass-failed -Wno-unused-variable "$@" -c msda.hip -o /tmp/abb_$name/msda.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../ab/lib_$name.so $(ls _obj/*.o | grep -v "/msda.o\|msda_exp.o\|probe_exp.o") /tmp/abb_$name/msda.o && echo "built $name"
