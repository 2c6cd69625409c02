#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_msda.py -m gpu -x -q -k "forced or variants" 2>&1 | tail -4
for v in 8 900; do timeout 120 python tools/msda_probe.py --shape micro --bs 2 --dir bwd --variant $v --iters 200 2>&1 | tail -1; done
for v in 32 900; do timeout 120 python tools/msda_probe.py --shape dec --bs 4 --lq 1100 --dir bwd --variant $v --iters 50 2>&1 | tail -1; done
for v in 8 900; do timeout 120 python tools/msda_probe.py --shape dec --bs 1 --lq 1100 --dir bwd --variant $v --iters 50 2>&1 | tail -1; done
