import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("SEMIDETR_EXPERIMENTS", "1")
import torch, bench
import semi_detr_amd as sda
import MultiScaleDeformableAttention as MSDA
lib = sda._lib.lib()
dev = torch.device("cuda:0")
wl_shapes = torch.as_tensor(bench.LEVELS, dtype=torch.long, device=dev)
v, sh, st, loc, attn, gout, Sx, Lx, lq = bench._msda_case(dev, bench.LEVELS, 4, 0, True)
gout = torch.rand_like(gout)
sda._lib.set_variant(0, int(sys.argv[1]) if len(sys.argv) > 1 else 73)
buf = (ctypes.c_ulonglong * 16)()
MSDA.ms_deform_attn_backward(v, sh, st, loc, attn, gout, 64); torch.cuda.synchronize()
lib.semidetr_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
MSDA.ms_deform_attn_backward(v, sh, st, loc, attn, gout, 64); torch.cuda.synchronize()
lib.semidetr_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
names = (["setup", "walk_tail_wait", "count", "scan", "fill", "walk", "geometry", "", "", "", "misses_n", ""] if len(sys.argv) > 1 and sys.argv[1] == "696"
         else ["enumerate", "sort", "walk", "misses", "tail", "flush", "", "", "rounds", "entries", "misses_n", "units"])
tot = sum(buf[i] for i in range(7))
for i, nme in enumerate(names):
    if nme: print(f"{nme:10s} {buf[i]:16d}" + (f"  {100.0*buf[i]/tot:5.1f}% of workgroup cycles" if i < 7 else ""))
if len(sys.argv) > 1 and sys.argv[1] == "696":
    print("rows flushed by sampling level 0..3:", [int(buf[12 + i]) for i in range(4)], "misses", int(buf[10]))
print("entries/round", buf[9] / max(1, buf[8]), "rounds/unit", buf[8] / max(1, buf[11]), "cycles/round", tot / max(1, buf[8]))
