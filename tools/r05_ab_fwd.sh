#!/bin/bash
# forward A/B of several product builds (ab/lib_<name>.so) on one box: event-timed probe (rotated inputs, REPS interleaved passes),
# min and median per (build, sigma)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; cp semi-detr_amd/csrc/libsemidetr_hip.so /tmp/lib_keep.so
rm -f /tmp/abf.txt
for rep in $(seq 1 ${REPS:-4}); do
for which in $LIBS; do
cp $R/ab/lib_$which.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
for sg in ${SIGMAS:-2.0}; do
SEMIDETR_EXPERIMENTS=0 timeout 300 python tools/msda_probe.py --shape enc --bs ${BS:-4} --dir ${DIR:-fwd} --iters ${ITERS:-60} --cold 6 --variant 0 --policy window --sigma $sg $EXTRA 2>&1 | grep "us  alg" | awk -v w=$which -v s=$sg '{print w, s, $6}' >> /tmp/abf.txt
done
done
done
cp /tmp/lib_keep.so $R/semi-detr_amd/csrc/libsemidetr_hip.so
python - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for l in open("/tmp/abf.txt"):
    w, s, us = l.split()
    d[(w, s)].append(float(us))
for k in sorted(d, key=lambda k: (k[1], k[0])):
    v = d[k]
    print("%-10s sigma %s  min %7.1f  median %7.1f  (%s)" % (k[0], k[1], min(v), statistics.median(v), " ".join("%.1f" % x for x in v)))
PY
