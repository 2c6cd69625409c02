#!/usr/bin/env python
"""Probe for profiling / tuning: run one MSDA shape `iters` times (forward and/or backward) with a forced
kernel variant and print event-timed microseconds per launch.
    python tools/msda_probe.py --shape enc --bs 4 --dir fwd --variant 1 --iters 20"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("SEMIDETR_EXPERIMENTS", "1")      # kernel variants live in the experiments build
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="enc", choices=["enc", "dec", "micro"])
    ap.add_argument("--bs", type=int, default=4)
    ap.add_argument("--lq", type=int, default=1100)
    ap.add_argument("--dir", default="both", choices=["fwd", "bwd", "both"])
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--fvariant", type=int, default=None)
    ap.add_argument("--sigma", type=float, default=2.0, help="encoder sample spread in pixels")
    ap.add_argument("--print-kernels", action="store_true", help="print what the library says it launched (KERNELS=...)")
    ap.add_argument("--policy", default="adaptive", choices=["adaptive", "patch", "window"],
                    help="encoder forward kernel (semidetr_msda_set_forward_policy)")
    ap.add_argument("--settle", type=int, default=0, help="synchronised warm-up launches (lets the adaptive forward policy see the data)")
    ap.add_argument("--cold", type=int, default=1, help="rotate this many distinct input sets (8 x 47 MB > the Infinity Cache)")
    ap.add_argument("--check", action="store_true", help="forward only: compare the forced variant's result with the patch kernel's")
    ap.add_argument("--io", default="locattn", choices=["locattn", "raw"],
                    help="raw (round 6): the fused-prologue entry points on bench.py's OWN inputs (bench.Workload: reference points, raw offsets "
                         "sigma 2 px, raw logits) -- what the headline step launches; shapes enc / dec only")
    ap.add_argument("--masked", action="store_true", help="with --io raw: the padding mask of bench.py's mixed-size batch in every call")
    a = ap.parse_args()
    if a.io == "raw":
        return main_raw(a)
    import semi_detr_amd as sda
    import MultiScaleDeformableAttention as MSDA
    sda._lib.set_variant(a.variant if a.fvariant is None else a.fvariant, a.variant)
    sda._lib.set_forward_policy(a.policy)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    S, M, D, L, P, LEVELS = bench.S, bench.M, bench.D, bench.L, bench.P, bench.LEVELS
    shapes = torch.as_tensor(LEVELS, dtype=torch.long, device=dev)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    n = a.bs
    value = torch.rand(n, S, M, D, device=dev) * 0.01
    if a.shape == "enc":
        lq = S
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h,
                                                    (torch.arange(w, device=dev) + 0.5) / w, indexing="ij"),
                                     -1).flip(-1).reshape(-1, 2) for h, w in LEVELS])
        inv = torch.tensor([[a.sigma / w, a.sigma / h] for h, w in LEVELS], device=dev).view(1, 1, 1, L, 1, 2)
        loc = (ref.view(1, S, 1, 1, 1, 2) + torch.randn(n, S, M, L, P, 2, device=dev) * inv).contiguous()
    else:
        lq = 300 if a.shape == "micro" else a.lq
        loc = torch.rand(n, lq, M, L, P, 2, device=dev)
    attn = torch.rand(n, lq, M, L, P, device=dev) + 1e-5
    attn = attn / attn.sum((-1, -2), keepdim=True)
    gout = torch.rand(n, lq, M * D, device=dev)
    sets = [(value, loc, attn, gout)]
    for _ in range(a.cold - 1):
        sets.append((torch.rand_like(value) * 0.01, loc.clone(), attn.clone(), gout.clone()))
    turn = [0]

    def pick():
        turn[0] += 1
        return sets[turn[0] % len(sets)]
    if a.check:
        out_v = MSDA.ms_deform_attn_forward(value, shapes, starts, loc, attn, 64)
        sda._lib.set_variant(0, 0)
        out_0 = MSDA.ms_deform_attn_forward(value, shapes, starts, loc, attn, 64)
        sda._lib.set_variant(a.variant if a.fvariant is None else a.fvariant, a.variant)
        print(f"CHECK max |variant - patch| = {(out_v - out_0).abs().max().item():.3e} (max |out| {out_0.abs().max().item():.3e})")
    runs = []
    if a.dir in ("fwd", "both"):
        def f():
            v, lo, at, _ = pick()
            MSDA.ms_deform_attn_forward(v, shapes, starts, lo, at, 64)
        runs.append(("fwd", f, False))
    if a.dir in ("bwd", "both"):
        def b_():
            v, lo, at, go = pick()
            MSDA.ms_deform_attn_backward(v, shapes, starts, lo, at, go, 64)
        runs.append(("bwd", b_, True))
    for name, fn, bw in runs:
        for _ in range(a.settle):
            fn()
            torch.cuda.synchronize()
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        if a.print_kernels:
            print("KERNELS=" + sda._lib.lib().semidetr_msda_last_kernels().decode())
        b = bench.msda_alg_bytes(n, lq, bw)
        print(f"{a.shape} bs{n} Lq{lq} {name} variant{a.variant}: {us:9.1f} us  alg {b/1e6:8.1f} MB  "
              f"{b/us/1e3:8.1f} GB/s  ({b/us/1e3/80:5.1f}% of 8 TB/s)")


def main_raw(a):
    """the fused entry points on the bench step's inputs (bench.Workload), same timing / printing as the reference-contract probe"""
    import semi_detr_amd as sda
    import MultiScaleDeformableAttention as MSDA
    assert a.shape in ("enc", "dec") and a.variant == 0 and a.sigma == 2.0, "--io raw: product kernels on bench.py's inputs (sigma 2 px)"
    sda._lib.set_forward_policy(a.policy)
    dev = torch.device("cuda:0")
    wl = bench.Workload(dev, 1234, recipe="coco10", io="raw", input_sets=max(1, a.cold), masked=a.masked)
    n, lq = a.bs, (wl.S if a.shape == "enc" else a.lq)
    mask = wl.mask[n] if a.masked else None
    turn = [0]

    def pick():
        turn[0] += 1
        r = turn[0]
        return wl.t[("value", n)][r % wl.rot], wl._args(a.shape, n, lq, r), wl.t[("enc_gout", n) if a.shape == "enc" else ("dec_gout", n, lq)][r % wl.rot]
    runs = []
    if a.dir in ("fwd", "both"):
        def f():
            v, args, _ = pick()
            MSDA.ms_deform_attn_fused_forward(v, wl.shapes, wl.starts, *args, mask)
        runs.append(("fwd", f, False))
    if a.dir in ("bwd", "both"):
        def b_():
            v, args, go = pick()
            MSDA.ms_deform_attn_fused_backward(v, wl.shapes, wl.starts, *args, go, mask)
        runs.append(("bwd", b_, True))
    for name, fn, bw in runs:
        for _ in range(max(a.settle, 6)):      # (synchronised: the adaptive policy and the backward's gather see the counts)
            fn()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.iters
        if a.print_kernels:
            print("KERNELS=" + sda._lib.lib().semidetr_msda_last_kernels().decode())
        b = bench.msda_alg_bytes(n, lq, bw)
        print(f"{a.shape} bs{n} Lq{lq} {name} raw{'+mask' if a.masked else ''}: {us:9.1f} us  alg {b/1e6:8.1f} MB  "
              f"{b/us/1e3:8.1f} GB/s  ({b/us/1e3/80:5.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
