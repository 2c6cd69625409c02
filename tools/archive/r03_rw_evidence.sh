#!/bin/bash
# Evidence for the region-window experiment (msda_rw.h, experiments build): parity, per-configuration times against the patch
# kernels on ONE box, the two ablations (no staging / no LDS loop), per-phase cycle counts.  -> gpurun_out/r03_rw_evidence.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
export SEMIDETR_EXPERIMENTS=1
cd $R
{
echo "# region-window LDS kernels (semi-detr_amd/csrc/msda_rw.h), forward variants 700-709, backward 7000-7009 (gather + region scatter)"
echo "# parity: tests/test_gpu_msda.py::test_encoder_self_attention_vs_oracle (all variants) and tests/test_gpu_fullsize.py -k encoder under 700,7000"
python -m pytest tests/test_gpu_msda.py -q -k "encoder_self_attention" 2>&1 | tail -1
SEMIDETR_TEST_VARIANT=700,7000 python -m pytest tests/test_gpu_fullsize.py -q -k "encoder" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
echo "# encoder shape bs 4 (N=4, Lq=S=22223), sigma 2 px; cfg: 0 = <512 thr, 8x16 region, margins 4/5>; 1 = <256,16x16,L0 global,4>; 2 = <512,16x16,L0 global,4>;"
echo "#   3 = <256,8x16,L0 global,3>; 4 = <256,8x8,L0 global,4>; 5 = <256,8x16,L0 global,4> (two workgroups per CU); 6 = <512,8x16,L0 global,5>;"
echo "#   8 / 9 = cfg 5 forward (cfg 0 gather) WITHOUT window staging / WITHOUT the LDS compute loop (results wrong, timing aids)"
for cfg in 0 1 2 3 4 5 6 8 9; do
    python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --fvariant $((700+cfg)) --variant $((7000+cfg)) --iters 20 2>&1 | grep "us  alg" | sed "s/^/[cfg $cfg] /"
done
python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --variant 0 --iters 20 2>&1 | grep "us  alg" | sed "s/^/[product kernels] /"
for sg in 1 4; do
  python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --fvariant 700 --variant 7000 --sigma $sg --iters 20 2>&1 | grep "us  alg" | sed "s/^/[cfg 0 sigma $sg] /"
  python $R/tools/msda_probe.py --shape enc --bs 4 --dir both --variant 0 --sigma $sg --iters 20 2>&1 | grep "us  alg" | sed "s/^/[product sigma $sg] /"
done
echo "# per-phase cycle counts (instrumented builds 707 / 7007: wave 0 of every workgroup; the laps themselves inflate the totals)"
python $R/tools/r03_rw_dbg.py 2>&1 | grep -v amdgpu.ids
} > $O/r03_rw_evidence.txt 2>&1
tail -50 $O/r03_rw_evidence.txt
