#!/bin/bash
# round 4: per-kernel averages of the fused-prologue step (--io raw) vs the reference-contract step under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for io in locattn raw; do
rm -rf $R/gpurun_out/rp_$io
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rp_$io -- python $R/bench.py --steps 8 --warmup 3 --no-micro --no-flavours --no-cpu-baseline --io $io > $R/gpurun_out/rp_$io.json 2> $R/gpurun_out/rp_$io.err
echo "== $io"; cd $R; python tools/summarize_prof.py bygrid gpurun_out/rp_$io | head -14 | cut -c1-120; cd /tmp
done
