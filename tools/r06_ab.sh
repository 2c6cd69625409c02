#!/bin/bash
# round 6: in-step A/B of product builds (ab/lib_<name>.so), then the GPU suite on the in-tree library.  LIBS="a b c" REPS=2 TESTS="tests/x.py ..."
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
EXTRA="${EXTRA:---no-flavours}" LIBS="$LIBS" REPS=${REPS:-2} bash tools/r05_ab_step.sh 2>&1 | grep "^\["
if [ -n "$TESTS" ]; then timeout -k 5 2400 python -m pytest $TESTS -x -q 2>&1 | tail -5; fi
