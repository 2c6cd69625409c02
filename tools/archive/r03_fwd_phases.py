#!/usr/bin/env python
"""Per-phase wave-0 cycle counts of the encoder forward (instrumented experiments build), reference contract vs fused prologue."""
import ctypes, os, sys
os.environ["SEMIDETR_EXPERIMENTS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import semi_detr_amd as sda
import MultiScaleDeformableAttention as MSDA
lib = sda._lib.lib()
buf = (ctypes.c_ulonglong * 16)()
dev = torch.device("cuda:0")
for io in ("locattn", "raw"):
    wl = bench.Workload(dev, 0, "coco10", io)
    v, a = wl.t[("value", 4)][0], wl._args("enc", 4, wl.S, 0)
    fn = MSDA.ms_deform_attn_fused_forward if io == "raw" else MSDA.ms_deform_attn_forward
    extra = () if io == "raw" else (64,)
    for _ in range(3):
        fn(v, wl.shapes, wl.starts, *a, *extra)
    torch.cuda.synchronize()
    lib.semidetr_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn(v, wl.shapes, wl.starts, *a, *extra)
    e1.record()
    torch.cuda.synchronize()
    lib.semidetr_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
    n = max(buf[11], 1)
    print(io, "us/launch %.1f" % (e0.elapsed_time(e1) * 100), "patches/launch", buf[11] // 10,
          "cycles per patch: wait %.0f records %.0f gather %.0f" % (buf[8] / n, buf[9] / n, buf[10] / n))
    go = wl.t[("enc_gout", 4)][0]
    fb = MSDA.ms_deform_attn_fused_backward if io == "raw" else MSDA.ms_deform_attn_backward
    for _ in range(3):
        fb(v, wl.shapes, wl.starts, *a, go, *extra)
    torch.cuda.synchronize()
    lib.semidetr_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
    e0.record()
    for _ in range(10):
        fb(v, wl.shapes, wl.starts, *a, go, *extra)
    e1.record()
    torch.cuda.synchronize()
    lib.semidetr_debug_counters(ctypes.cast(buf, ctypes.c_void_p), 1)
    n = max(buf[15], 1)
    print(io, "bwd us/launch %.1f" % (e0.elapsed_time(e1) * 100), "gather patches sampled/launch", buf[15] // 10,
          "cycles per patch: records %.0f gather %.0f stores %.0f" % (buf[12] / n, buf[13] / n, buf[14] / n))
    del wl
