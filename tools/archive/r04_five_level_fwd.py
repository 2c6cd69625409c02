"""Encoder self-attention FORWARD on the five-level (COCO-Full) pyramid, bs 4, rotated inputs: patch kernel vs region-window
configurations of the experiments build (forward variants).   python tools/r04_five_level_fwd.py 0 700 715 718"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SEMIDETR_EXPERIMENTS", "1")
import torch
import semi_detr_amd as sda
import MultiScaleDeformableAttention as MSDA
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, D, P, n = 8, 32, 4, 4
levels = [(100, 167), (50, 84), (25, 42), (13, 21), (7, 11)]
L = len(levels)
shapes = torch.as_tensor(levels, dtype=torch.long, device=dev)
starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
S = int((shapes[:, 0] * shapes[:, 1]).sum())
ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w,
                                            indexing="ij"), -1).flip(-1).reshape(-1, 2) for h, w in levels])
sda._lib.set_forward_policy("patch")
for sigma in [float(x) for x in os.environ.get("SIGMAS", "1.0 2.0 3.0").split()]:
    inv = torch.tensor([[sigma / w, sigma / h] for h, w in levels], device=dev).view(1, 1, 1, L, 1, 2)
    sets = []
    for _ in range(5):
        a = torch.rand(n, S, M, L, P, device=dev) + 1e-5
        sets.append((torch.rand(n, S, M, D, device=dev) * 0.01,
                     (ref.view(1, S, 1, 1, 1, 2) + torch.randn(n, S, M, L, P, 2, device=dev) * inv).contiguous(),
                     (a / a.sum((-1, -2), keepdim=True)).contiguous()))
    want = None
    for v in [int(x) for x in sys.argv[1:]] or [0]:
        sda._lib.set_variant(v, 0)
        outs = []
        for i in range(3):
            outs.append(MSDA.ms_deform_attn_forward(sets[i][0], shapes, starts, sets[i][1], sets[i][2], 64))
        if want is None:
            want = [o.clone() for o in outs]
        err = max(float((o - w).abs().max()) for o, w in zip(outs, want))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for i in range(20):
            MSDA.ms_deform_attn_forward(sets[i % 5][0], shapes, starts, sets[i % 5][1], sets[i % 5][2], 64)
        e1.record(); torch.cuda.synchronize()
        print(f"five levels S={S} bs {n} sigma {sigma} fwd variant {v}: {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us  max|diff vs variant {sys.argv[1] if len(sys.argv) > 1 else 0}| {err:.2e}  {sda._lib.lib().semidetr_msda_last_kernels().decode()}")
