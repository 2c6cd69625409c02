#!/bin/bash
# round 6: per-kernel rocprofv3 averages of the encoder backward / forward for product builds ab/lib_<name>.so.  LIBS=... DIR=bwd|fwd IO=locattn|raw
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
LIBS="$LIBS" POLICY=${POLICY:-window} DIR=${DIR:-bwd} BS=${BS:-4} SIGMA=${SIGMA:-2.0} bash tools/r05_ab_kern.sh 2>&1 | grep "msda_\|us  alg"
