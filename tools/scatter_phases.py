"""Per-phase wave-0 cycles of the instrumented region kernels (experiments build): 696 = region scatter, 6901 = fused
encoder backward.   python tools/scatter_phases.py 6901 [bs]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SEMIDETR_EXPERIMENTS", "1")
import torch, bench
import semi_detr_amd as sda
import MultiScaleDeformableAttention as MSDA
lib = sda._lib.lib()
dev = torch.device("cuda:0")
var = int(sys.argv[1]) if len(sys.argv) > 1 else 6901
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
v, sh, st, loc, attn, gout, Sx, Lx, lq = bench._msda_case(dev, bench.LEVELS, bs, 0, True)
gout = torch.rand_like(gout)
sda._lib.set_variant(0, var)
buf = (ctypes.c_ulonglong * 16)()
for _ in range(2):
    MSDA.ms_deform_attn_backward(v, sh, st, loc, attn, gout, 64); torch.cuda.synchronize()
    lib.semidetr_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
names = ["setup", "walk_tail_wait", "count", "scan", "fill", "walk", "geometry", "dot", "combine"]
tot = sum(buf[i] for i in range(9))
for i, nme in enumerate(names):
    print(f"{nme:15s} {buf[i]:16d}  {100.0*buf[i]/max(tot,1):5.1f}% of wave-0 cycles")
print("rows flushed by sampling level 0..3:", [int(buf[12 + i]) for i in range(4)], "misses", int(buf[10]))
