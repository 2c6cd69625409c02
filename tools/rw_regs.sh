#!/bin/bash
# quick register / spill check of ONE region-window configuration (seconds instead of the whole library):
#   tools/rw_regs.sh "LocAttnIO, 768, 16, 16, -1, 6, 4, false, 0, 20" [kernel name, default msda_rw_d32]
cd /root/repo
mkdir -p /tmp/asm
K=${2:-msda_rw_d32}
N=$(grep -n '#include "msda_rw.h"' semi-detr_amd/csrc/msda.hip | head -1 | cut -d: -f1)
{ sed -n "1,${N}p" semi-detr_amd/csrc/msda.hip | grep -v "msda_fast_experiments.h"; echo "}"; echo "void *force_it() { return (void *)&$K<$1>; }"; } > /tmp/asm/rwtest.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Iinclude -Isemi-detr_amd/csrc -Wno-pass-failed -Wno-unused-variable -Wno-unused-function \
  --offload-device-only -S /tmp/asm/rwtest.hip -o /tmp/asm/rwtest.s 2>&1 | grep "error" | head -20
grep "vgpr_count\|vgpr_spill\|sgpr_count\|sgpr_spill\|private_segment_fixed" /tmp/asm/rwtest.s | tr -s ' ' | tr '\n' ' '; echo
