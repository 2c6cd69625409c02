"""Timing of encoder self-attention on the five-level (COCO-Full) pyramid next to the four-level one, bs 4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import semi_detr_amd  # noqa: F401
import MultiScaleDeformableAttention as MSDA
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, D, P, n = 8, 32, 4, 4
for levels in ([(100, 167), (50, 84), (25, 42), (13, 21)], [(100, 167), (50, 84), (25, 42), (13, 21), (7, 11)]):
    L = len(levels)
    shapes = torch.as_tensor(levels, dtype=torch.long, device=dev)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = torch.rand(n, S, M, D, device=dev) * 0.01
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w,
                                                indexing="ij"), -1).flip(-1).reshape(-1, 2) for h, w in levels])
    inv = torch.tensor([[2.0 / w, 2.0 / h] for h, w in levels], device=dev).view(1, 1, 1, L, 1, 2)
    loc = (ref.view(1, S, 1, 1, 1, 2) + torch.randn(n, S, M, L, P, 2, device=dev) * inv).contiguous()
    attn = torch.rand(n, S, M, L, P, device=dev) + 1e-5
    attn = attn / attn.sum((-1, -2), keepdim=True)
    gout = torch.rand(n, S, M * D, device=dev)
    for name, fn in (("fwd", lambda: MSDA.ms_deform_attn_forward(value, shapes, starts, loc, attn, 64)),
                     ("bwd", lambda: MSDA.ms_deform_attn_backward(value, shapes, starts, loc, attn, gout, 64))):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        print(f"{L} levels, S = {S}, bs {n}, encoder {name}: {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us   kernels {semi_detr_amd._lib.lib().semidetr_msda_last_kernels().decode()}")
