#!/bin/bash
# phase breakdown of the instrumented product window forward (750) + timing of the product
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
python tools/rw_phases.py 750 2>&1 | tail -14
python tools/msda_probe.py --shape enc --bs 4 --dir fwd --policy window --cold 6 --iters 30 2>&1 | tail -1
