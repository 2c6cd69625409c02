"""Per-phase cycle counts of the instrumented region-window kernel (forward variant 707 / backward 7007)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tools/ -> repo root
os.environ.setdefault("SEMIDETR_EXPERIMENTS", "1")
import torch, bench
import semi_detr_amd as sda
import MultiScaleDeformableAttention as MSDA
lib = sda._lib.lib()
dev = torch.device("cuda:0")
v, sh, st, loc, attn, gout, Sx, Lx, lq = bench._msda_case(dev, bench.LEVELS, 4, 0, True)
gout = torch.rand_like(gout)
names = ["setup", "stage_issue", "boundary_wait", "stage_store", "store_wait", "geometry", "next_loads", "compute", "outside", "results"]
fv = int(sys.argv[1]) if len(sys.argv) > 1 else 707
for which in (("fwd", "bwd") if fv == 707 else ("fwd",)):
    sda._lib.set_variant(fv, 7007 if fv == 707 else 0)
    buf = (ctypes.c_ulonglong * 16)()
    run = (lambda: MSDA.ms_deform_attn_forward(v, sh, st, loc, attn, 64)) if which == "fwd" else \
          (lambda: MSDA.ms_deform_attn_backward(v, sh, st, loc, attn, gout, 64))
    run(); torch.cuda.synchronize()
    lib.semidetr_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
    run(); torch.cuda.synchronize()
    lib.semidetr_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
    tot = sum(buf[i] for i in range(10))
    print(f"--- {which}: wave 0 of every workgroup, cycles by phase; rounds {buf[10]}, regions {buf[11]}, cycles/region {tot / max(1, buf[11]):.0f}")
    for i, nme in enumerate(names):
        print(f"{nme:14s} {buf[i]:16d}  {100.0 * buf[i] / tot:5.1f} %   {buf[i] / max(1, buf[11]):9.0f} per region")
