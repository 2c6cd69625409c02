#!/bin/bash
# round 4: where the region-window forward (702) stops paying: sigma sweep, rotated inputs (6 sets), two rounds
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for sg in 1.5 2.0 2.5 3.0 3.5; do for v in 0 702; do
timeout 120 python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 0 --fvariant $v --iters 24 --sigma $sg --cold 6 2>&1 | tail -1 | sed "s/^/[sigma $sg fwd $v] /"
done; done; done
for sg in 1.5 2.0 2.5 3.0 3.5; do for v in 0 702; do
timeout 120 python tools/msda_probe.py --shape enc --bs 1 --dir fwd --variant 0 --fvariant $v --iters 24 --sigma $sg --cold 6 2>&1 | tail -1 | sed "s/^/[bs1 sigma $sg fwd $v] /"
done; done
python - <<'PY'
import math
# P(|N(0, s)| > R) per axis -> fraction of samples further than R px from their query's centre on either axis
for s in (1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0):
    for R in (4.0, 4.5):
        p = math.erfc(R / s / math.sqrt(2))
        print("sigma %.1f  R %.1f: per axis %.3f, either axis %.3f" % (s, R, p, 1 - (1 - p) ** 2))
PY
