"""How much does an out-of-range (kOob -> hardware zero) corner load cost?  Encoder forward bs 4 with a growing share of
samples pushed outside the map: if the time falls in proportion, folding duplicate corners of a query's points into one
load + zero-offset loads would cut the L1 traffic of the forward."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import semi_detr_amd  # noqa: F401
import MultiScaleDeformableAttention as MSDA
dev = torch.device("cuda:0")
v, sh, st, loc, attn, gout, Sx, Lx, lq = bench._msda_case(dev, bench.LEVELS, 4, 0, True)
for frac in (0.0, 0.25, 0.5, 1.0):
    lo = loc.clone()
    mask = torch.rand(lo.shape[:-1], device=dev) < frac
    lo[mask] = 5.0                     # far outside: sample skipped, all four corners out of range
    lo = lo.contiguous()
    for _ in range(3):
        MSDA.ms_deform_attn_forward(v, sh, st, lo, attn, 64)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        MSDA.ms_deform_attn_forward(v, sh, st, lo, attn, 64)
    e1.record(); torch.cuda.synchronize()
    print(f"fraction of samples out of range {frac:4.2f}: {e0.elapsed_time(e1) * 1e3 / 20:7.1f} us")
