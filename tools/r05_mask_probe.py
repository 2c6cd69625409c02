"""Encoder forward (fused prologue, bs 4, 800x1333 pyramid, sigma 2 px) with / without a padding mask, event-timed over six
rotating input sets: which part of the mask handling costs what (tools/r05_mask_ab.sh swaps builds of the product library)."""
import sys
import torch
sys.path.insert(0, ".")
import bench  # noqa: E402
import semi_detr_amd as sda  # noqa: E402
import MultiScaleDeformableAttention as MSDA  # noqa: E402

dev = torch.device("cuda", 0)
policy = sys.argv[1] if len(sys.argv) > 1 else "window"
sda._lib.set_forward_policy(policy)
res = {}
for tag, masked in (("nomask", False), ("mixed", True)):
    wl = bench.Workload(dev, 1234, "coco10", "raw", masked=masked)
    for n in (4, 1):
        sets = [(wl.t[("value", n)][r], wl._args("enc", n, wl.S, r)) for r in range(wl.rot)]
        masks = [None] if not masked else [wl.mask[n]]
        if not masked:
            masks.append(torch.zeros(n, wl.S, dtype=torch.bool, device=dev))
        for mk in masks:
            name = "%s%s_bs%d" % (tag, "" if mk is None or masked else "_allfalse", n)
            for v, a in sets:
                MSDA.ms_deform_attn_fused_forward(v, wl.shapes, wl.starts, *a, mk)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                for v, a in sets:
                    MSDA.ms_deform_attn_fused_forward(v, wl.shapes, wl.starts, *a, mk)
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) * 1e3 / (8 * len(sets))
    del wl
    torch.cuda.empty_cache()
print(" ".join("%s %.1f" % kv for kv in res.items()), sda._lib.lib().semidetr_msda_last_kernels().decode())
