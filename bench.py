#!/usr/bin/env python
"""bench.py -- throughput of the Semi-DETR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE pass of the hot path of a Semi-DETR teacher-student training iteration for one GPU's share
of the batch (configs/detr_ssod recipe: 1 labeled + 4 unlabeled images per GPU, 800x1333, DINO-R50 shapes;
SURVEY.md section 3.1), on synthetic inputs already resident in HBM:

    EMA teacher update (47 M fp32 params, ~500 tensors)                              [mean_teacher.py:37-64]
    60 MSDA forward launches  (6 enc + 6 dec layers x {sup bs1, teacher bs4, student-nograd bs4,
                               student forward_dummy bs4, teacher forward_dummy bs4})
    teacher pseudo labels: box decoding + class-aware NMS (4 x 900 x 80) + mean/std filter + weak->strong box warp
                                                        [dino_detr_ssod_head.py:1364-1395, dino_detr_ssod.py:918-939]
    39 Hungarian matchings in 3 batched calls (4 inline unsup + 7x1 sup + 7x4 unsup)  [hungarian_assigner.py]
    24 MSDA backward launches (sup bs1 + student forward_dummy bs4, 6 enc + 6 dec each)
    N > 1: mean all-reduce of the 60 M-float gradient arena over RCCL/xGMI, bucketed, overlapped with backward

The dense parts of the model (ResNet-50, Linear/FFN GEMMs) are NOT in the step: this run measures the hot
path the north star names, not the whole detector.  `value` = images/s of that path over all ranks.
Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

LEVELS = [(100, 167), (50, 84), (25, 42), (13, 21)]        # 800x1333 -> C3..C5 + stride-2 level
S = sum(h * w for h, w in LEVELS)                           # 22223
M, D, L, P = 8, 32, 4, 4
NUM_QUERY, DN_PAD = 900, 200
IMAGES_PER_GPU = 5                                          # 1 labeled + 4 unlabeled (two views each)
GRAD_ELEMS = 60_000_000                                     # student DINO-R50 + projector
HBM_PEAK_GBS = 8000.0                                       # MI355X_MICROARCH.md: 8 TB/s spec


def msda_alg_bytes(N, Lq, backward):
    """SURVEY.md section 8(d) algorithmic bytes of one MSDA launch (fp32)."""
    e = 4
    K = N * Lq * M * L * P
    vmap = N * S * M * D * e
    g4 = 4 * K * D * e
    vt = min(vmap, g4)
    bl, ba, bo = K * 2 * e, K * e, N * Lq * M * D * e
    fwd = vt + bl + ba + bo
    bwd = bo + vt + bl + ba + bl + ba + vmap + vt
    return bwd if backward else fwd


def dino_param_sizes():
    """Parameter tensor sizes of a DINO-DETR R50 detector (ResNet-50 without fc + 6/6-layer deformable
    transformer + heads), ~47 M elements in ~500 tensors -- the list MeanTeacher walks every step."""
    sizes = [64 * 3 * 49, 64, 64]
    inp = 64
    for planes, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
        for b in range(blocks):
            sizes += [planes * inp, planes, planes, planes * planes * 9, planes, planes, planes * 4 * planes,
                      planes * 4, planes * 4]
            if b == 0:
                sizes += [planes * 4 * inp, planes * 4, planes * 4]
            inp = planes * 4
    d, ff = 256, 2048
    msda = [d * 256, 256, d * 128, 128, d * d, d, d * d, d]
    ffn = [d * ff, ff, ff * d, d, d, d, d, d]
    for _ in range(6):
        sizes += msda + ffn                                               # encoder layer
    for _ in range(6):
        sizes += msda + ffn + [3 * d * d, 3 * d, d * d, d, d, d]          # decoder layer (+ self-attn)
    for c in (512, 1024, 2048):
        sizes += [c * d, d, d, d]                                          # input_proj
    sizes += [2048 * d * 9, d, d, d, 4 * d, NUM_QUERY * d, 100 * d, d * d, d, 2 * d * d, d]
    for _ in range(7):
        sizes += [d * 80, 80, d * d, d, d * d, d, d * 4, 4]                # cls / reg branches
    return sizes


class Workload:
    def __init__(self, dev, seed):
        import semi_detr_amd as sda
        self.sda, self.dev = sda, dev
        g = torch.Generator(device=dev).manual_seed(seed)
        self.shapes = torch.as_tensor(LEVELS, dtype=torch.long, device=dev)
        self.starts = torch.cat([self.shapes.new_zeros(1), (self.shapes[:, 0] * self.shapes[:, 1]).cumsum(0)[:-1]])
        # encoder: query = every pixel, samples around its own centre (sigma = 2 px of that level)
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h,
                                                    (torch.arange(w, device=dev) + 0.5) / w, indexing="ij"),
                                     -1).flip(-1).reshape(-1, 2) for h, w in LEVELS])          # (S, 2) x,y
        inv = torch.tensor([[2.0 / w, 2.0 / h] for h, w in LEVELS], device=dev).view(1, 1, 1, L, 1, 2)

        def rand(*s):
            return torch.rand(*s, generator=g, device=dev)

        def randn(*s):
            return torch.randn(*s, generator=g, device=dev)

        def attn(n, lq):
            a = rand(n, lq, M, L, P) + 1e-5
            return (a / a.sum((-1, -2), keepdim=True)).contiguous()

        self.t = {}
        for n in (1, 4):
            self.t[("value", n)] = rand(n, S, M, D) * 0.01
            self.t[("enc_loc", n)] = (ref.view(1, S, 1, 1, 1, 2) + randn(n, S, M, L, P, 2) * inv).contiguous()
            self.t[("enc_attn", n)] = attn(n, S)
            self.t[("enc_gout", n)] = rand(n, S, M * D)
            for lq in (NUM_QUERY, NUM_QUERY + DN_PAD):
                # decoder: reference boxes anywhere, offsets scaled by the box size
                c = rand(n, lq, 1, 1, 1, 2)
                wh = rand(n, lq, 1, 1, 1, 2) * 0.3 + 0.02
                self.t[("dec_loc", n, lq)] = (c + randn(n, lq, M, L, P, 2) * wh * 0.5).contiguous()
                self.t[("dec_attn", n, lq)] = attn(n, lq)
                self.t[("dec_gout", n, lq)] = rand(n, lq, M * D)
        # matcher inputs: 7 layers x images, G ~ U{1..15}
        rng = np.random.default_rng(seed)
        self.asg = sda.HungarianAssigner(cls_cost=dict(type="FocalLossCost", weight=2.0),
                                         reg_cost=dict(type="BBoxL1Cost", weight=5.0, box_format="xywh"),
                                         iou_cost=dict(type="IoUCost", iou_mode="giou", weight=2.0))

        def problems(n_img, layers):
            gts, labs, metas = [], [], []
            for _ in range(n_img):
                G = int(rng.integers(1, 16))
                xy = torch.rand(G, 2, generator=g, device=dev) * torch.tensor([1000.0, 560.0], device=dev)
                wh = torch.rand(G, 2, generator=g, device=dev) * torch.tensor([300.0, 220.0], device=dev) + 16
                gts.append(torch.cat([xy, xy + wh], -1))
                labs.append(torch.randint(0, 80, (G,), generator=g, device=dev))
                metas.append(dict(img_shape=(800, 1333, 3)))
            B = n_img * layers
            bp = torch.cat([rand(B, NUM_QUERY, 2), rand(B, NUM_QUERY, 2) * 0.5 + 0.01], -1)
            cp = randn(B, NUM_QUERY, 80) * 3
            return bp, cp, gts * layers, labs * layers, metas * layers

        self.match_sets = [problems(4, 1), problems(1, 7), problems(4, 7)]
        # teacher outputs of the 4 unlabeled images (last decoder layer): mostly background, clustered boxes
        self.t_logits = (randn(4, NUM_QUERY, 80) * 2.0 - 5.0).contiguous()
        k = NUM_QUERY // 8
        cxcy, wh = rand(4, NUM_QUERY, 2), rand(4, NUM_QUERY, 2) * 0.3 + 0.02
        src = torch.randint(0, k, (NUM_QUERY - k,), generator=g, device=dev)
        cxcy[:, k:] = cxcy[:, src] + randn(4, NUM_QUERY - k, 2) * 0.01
        wh[:, k:] = wh[:, src] * (1 + randn(4, NUM_QUERY - k, 2) * 0.05)
        self.t_boxes = torch.cat([cxcy, wh], -1).contiguous()
        self.t_metas = [dict(img_shape=(800, 1333, 3))] * 4
        self.warp = [torch.tensor([[-0.9, 0.0, 1200.0], [0.0, 0.9, 12.0], [0.0, 0.0, 1.0]], device=dev)] * 4
        sizes = dino_param_sizes()
        self.n_params = sum(sizes)
        self.teacher = [randn(n) for n in sizes]
        self.student = [randn(n) for n in sizes]
        self.groups = {}
        self.events = []

    # -- group timing with events on the launch stream (torch's current stream is the stream we pass down)
    def _timed(self, name, launches, nbytes, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        if self.record:
            self.events.append((name, launches, nbytes, e0, e1))

    def _fwd(self, kind, n, lq, reps):
        import MultiScaleDeformableAttention as MSDA
        v = self.t[("value", n)]
        loc = self.t[("enc_loc", n)] if kind == "enc" else self.t[("dec_loc", n, lq)]
        a = self.t[("enc_attn", n)] if kind == "enc" else self.t[("dec_attn", n, lq)]

        def run():
            for _ in range(reps):
                MSDA.ms_deform_attn_forward(v, self.shapes, self.starts, loc, a, 64)
        self._timed(f"msda_fwd_{kind}_bs{n}_Lq{lq}", reps, msda_alg_bytes(n, lq, False), run)

    def _bwd(self, kind, n, lq, reps):
        import MultiScaleDeformableAttention as MSDA
        v = self.t[("value", n)]
        loc = self.t[("enc_loc", n)] if kind == "enc" else self.t[("dec_loc", n, lq)]
        a = self.t[("enc_attn", n)] if kind == "enc" else self.t[("dec_attn", n, lq)]
        go = self.t[("enc_gout", n)] if kind == "enc" else self.t[("dec_gout", n, lq)]

        def run():
            for _ in range(reps):
                MSDA.ms_deform_attn_backward(v, self.shapes, self.starts, loc, a, go, 64)
        self._timed(f"msda_bwd_{kind}_bs{n}_Lq{lq}", reps, msda_alg_bytes(n, lq, True), run)

    def _pseudo(self):
        # queued behind the teacher's forward; the lists are only needed where the unsupervised loss matches them
        self.pending = self.sda.teacher_pseudo_labels(self.t_logits, self.t_boxes, self.t_metas, wait=False)

    def _pseudo_finish(self):
        boxes, labels, scores = self.pending.result()
        self.sda.transform_bboxes(boxes, self.warp, [m["img_shape"] for m in self.t_metas])

    def _match(self, i):
        bp, cp, gts, labs, metas = self.match_sets[i]
        self._timed("hungarian_batch", 1, 0,
                    lambda: self.asg.assign_batch(bp, cp, gts, labs, metas, check=False))

    def step(self, reducer=None, record=False):
        self.record = record
        sda = self.sda
        if reducer is not None:
            reducer.start()
        self._timed("ema", 1, 12 * self.n_params, lambda: sda.ema_update_(self.teacher, self.student, 0.999))
        q, qd = NUM_QUERY, NUM_QUERY + DN_PAD
        self._fwd("enc", 1, S, 6); self._fwd("dec", 1, qd, 6)            # supervised student forward
        self._fwd("enc", 4, S, 6); self._fwd("dec", 4, q, 6)             # teacher simple_test
        self._timed("pseudo_label", 1, 0, self._pseudo)
        self._fwd("enc", 4, S, 6); self._fwd("dec", 4, q, 6)             # student no-grad forward
        self._timed("pseudo_label", 0, 0, self._pseudo_finish)           # lists + weak->strong warp (compute_pseudo_label_loss)
        self._match(0)                                                   # inline matching, unsup_loss
        self._fwd("enc", 4, S, 6); self._fwd("dec", 4, qd, 6)            # student forward_dummy
        self._fwd("enc", 4, S, 6); self._fwd("dec", 4, qd, 6)            # teacher forward_dummy
        self._match(1); self._match(2)                                   # sup + unsup loss()
        self._bwd("dec", 4, qd, 6)
        if reducer is not None:
            reducer.launch_ready(0.25)
        self._bwd("enc", 4, S, 6)
        if reducer is not None:
            reducer.launch_ready(0.6)
        self._bwd("dec", 1, qd, 6); self._bwd("enc", 1, S, 6)
        if reducer is not None:
            reducer.finish()

    def group_stats(self):
        torch.cuda.synchronize()
        agg = {}
        for name, launches, nbytes, e0, e1 in self.events:
            a = agg.setdefault(name, dict(ms=0.0, launches=0, bytes=nbytes))
            a["ms"] += e0.elapsed_time(e1)
            a["launches"] += launches
        return agg


def _graph_time(fn, per_graph=20, reps=9, replays=5):
    """Device time per call: the launches captured once in a HIP graph and replayed (what a training step sees --
    launches are queued ahead of the GPU).  Returns (median, p10, p90) in microseconds over `reps` timings."""
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(graph):
        for _ in range(per_graph):
            fn()
    graph.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (replays * per_graph))
    del graph
    return float(np.median(ts)), float(np.percentile(ts, 10)), float(np.percentile(ts, 90))


def hbm_stream_peak(dev):
    """Measured streaming rate of this GPU (device-to-device copy of 1 GiB: read + write), GB/s."""
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2 * a.numel() * 4 * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def _msda_case(dev, levels, N, Lq, encoder):
    shapes = torch.as_tensor(levels, dtype=torch.long, device=dev)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    Sx, Lx = int((shapes[:, 0] * shapes[:, 1]).sum()), len(levels)
    value = torch.rand(N, Sx, M, D, device=dev) * 0.01
    if encoder:      # reference-point-centred: pixel centre + N(0, (2 px)^2) on every level
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h,
                                                    (torch.arange(w, device=dev) + 0.5) / w, indexing="ij"), -1)
                         .flip(-1).reshape(-1, 2) for h, w in levels])
        wh = torch.tensor([[w, h] for h, w in levels], dtype=torch.float32, device=dev)
        loc = ref.view(1, Sx, 1, 1, 1, 2) + torch.randn(N, Sx, M, Lx, P, 2, device=dev) * (2.0 / wh).view(1, 1, 1, Lx, 1, 2)
        Lq = Sx
    else:
        loc = torch.rand(N, Lq, M, Lx, P, 2, device=dev)
    attn = torch.rand(N, Lq, M, Lx, P, device=dev) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    gout = torch.ones(N, Lq, M * D, device=dev)
    return value, shapes, starts, loc.contiguous(), attn, gout, Sx, Lx, Lq


def _alg_bytes(N, Sx, Lx, Lq, backward):
    e = 4
    K = N * Lq * M * Lx * P
    vmap, g4 = N * Sx * M * D * e, 4 * K * D * e
    vt, bl, ba, bo = min(vmap, g4), K * 2 * e, K * e, N * Lq * M * D * e
    return (bo + vt + bl + ba + bl + ba + vmap + vt) if backward else (vt + bl + ba + bo)


def microbench(dev, iters=200, warm=20):
    """BASELINE.json metric shape: N=2, Lq=300, L=4, M=8, P=4, D=32, S=22223; test.py input distributions
    (seed 3).  Plus the secondary shapes SURVEY.md section 8(d) names, each fwd / bwd as median [p10, p90]."""
    import MultiScaleDeformableAttention as MSDA
    torch.manual_seed(3)
    N, Lq = 2, 300
    value, shapes, starts, loc, attn, gout, _, _, _ = _msda_case(dev, LEVELS, N, Lq, False)
    res = {}
    for name, fn in (("fwd", lambda: MSDA.ms_deform_attn_forward(value, shapes, starts, loc, attn, 64)),
                     ("bwd", lambda: MSDA.ms_deform_attn_backward(value, shapes, starts, loc, attn, gout, 64))):
        for _ in range(warm):
            fn()
        # (a) eager launches from Python: includes the host cost of each call (allocation + ctypes), which at
        #     this size is larger than the forward kernel itself
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters // 5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / (iters // 5))
        res[name + "_eager_us"] = float(np.median(ts))
        # (b) graph replay: device time per call
        res[name + "_us"], res[name + "_p10_us"], res[name + "_p90_us"] = _graph_time(fn)
    res["fwd_bwd_us"] = res["fwd_us"] + res["bwd_us"]
    b = msda_alg_bytes(N, Lq, False) + msda_alg_bytes(N, Lq, True)
    res["alg_bytes"] = b
    res["frac_hbm_peak"] = b / (res["fwd_bwd_us"] * 1e-6) / (HBM_PEAK_GBS * 1e9)
    peak = hbm_stream_peak(dev)
    res["hbm_stream_measured_gbs"] = peak
    res["frac_hbm_measured"] = b / (res["fwd_bwd_us"] * 1e-6) / (peak * 1e9)
    sec = {}
    five = LEVELS + [(7, 11)]
    for key, levels, n, lq, enc in (("decoder_bs2_Lq1100", LEVELS, 2, 1100, False), ("encoder_bs2_Lq22223", LEVELS, 2, 0, True),
                                    ("five_level_bs2_Lq900", five, 2, 900, False)):
        v, sh, st, lo, at, go, Sx, Lx, lq = _msda_case(dev, levels, n, lq, enc)
        f = lambda: MSDA.ms_deform_attn_forward(v, sh, st, lo, at, 64)           # noqa: E731
        bw = lambda: MSDA.ms_deform_attn_backward(v, sh, st, lo, at, go, 64)     # noqa: E731
        f(); bw()
        tf, tb = _graph_time(f, reps=5), _graph_time(bw, reps=5)
        bf, bb = _alg_bytes(n, Sx, Lx, lq, False), _alg_bytes(n, Sx, Lx, lq, True)
        g4 = 4 * n * lq * M * Lx * P * D * 4
        sec[key] = {"fwd_us": tf[0], "fwd_p10_p90_us": [tf[1], tf[2]], "bwd_us": tb[0], "bwd_p10_p90_us": [tb[1], tb[2]],
                    "fwd_alg_gbs": bf / (tf[0] * 1e-6) / 1e9, "bwd_alg_gbs": bb / (tb[0] * 1e-6) / 1e9,
                    "fwd_gather_rate_gbs": g4 / (tf[0] * 1e-6) / 1e9}
        del v, lo, at, go
    res["secondary_shapes"] = sec
    return res


def module_bench(dev, iters=10):
    """Next-row evidence: one encoder MSDeformAttn layer (bs 4, 800x1333, d_model
    256) forward + backward with the prologue/epilogue fused into the sampling kernels vs the reference's
    op-by-op sequence (torch softmax / location arithmetic around MSDeformAttnFunction).  Includes the four
    Linear layers (hipBLASLt through torch) in both cases."""
    from semi_detr_amd import MSDeformAttn
    torch.manual_seed(0)
    m = MSDeformAttn(256, L, M, P).to(dev)
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.01)
        m.attention_weights.weight.normal_(0, 0.05)
    shapes = torch.as_tensor(LEVELS, dtype=torch.long, device=dev)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    n = 4
    src = torch.randn(n, S, 256, device=dev, requires_grad=True)
    pos = torch.randn(n, S, 256, device=dev)
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h,
                                                (torch.arange(w, device=dev) + 0.5) / w, indexing="ij"), -1)
                     .flip(-1).reshape(-1, 2) for h, w in LEVELS])
    ref = ref.view(1, S, 1, 2).expand(n, S, L, 2).contiguous()
    res = {}
    for fused in (True, False):
        m.fuse_prologue = fused
        for _ in range(2):
            m(src + pos, ref, src, shapes, starts).sum().backward()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            m(src + pos, ref, src, shapes, starts).sum().backward()
        e1.record()
        torch.cuda.synchronize()
        res["fused_ms" if fused else "op_by_op_ms"] = e0.elapsed_time(e1) / iters
    res["speedup"] = res["op_by_op_ms"] / res["fused_ms"]
    res["what"] = "MSDeformAttn encoder layer fwd+bwd, bs4, Lq=S=22223, d_model 256, incl. Linear layers"
    return res


def warmup_stage_bench(dev, iters=20):
    """Next-row evidence: the warm-up stage's matcher and classification loss for one
    loss() call of the SSOD step -- 7 layers x 5 images = 35 one-to-many assignment problems (Q=900, G~U[1,15]) in
    one launch, and the fused task-aligned focal loss (sigmoid + loss + gradient) over the 35 x 900 x 80 logits."""
    import semi_detr_amd as sda
    g = torch.Generator(device=dev).manual_seed(7)
    rng = np.random.default_rng(7)
    B, Q, C = 35, NUM_QUERY, 80
    bp = torch.cat([torch.rand(B, Q, 2, generator=g, device=dev), torch.rand(B, Q, 2, generator=g, device=dev) * 0.3 + 0.02], -1)
    logits = torch.randn(B, Q, C, generator=g, device=dev) * 2 - 3
    gts, labs, metas = [], [], []
    for _ in range(B):
        G = int(rng.integers(1, 16))
        xy = torch.rand(G, 2, generator=g, device=dev) * torch.tensor([1000.0, 560.0], device=dev)
        wh = torch.rand(G, 2, generator=g, device=dev) * torch.tensor([300.0, 220.0], device=dev) + 16
        gts.append(torch.cat([xy, xy + wh], -1))
        labs.append(torch.randint(0, C, (G,), generator=g, device=dev))
        metas.append(dict(img_shape=(800, 1333, 3)))
    asg, crit = sda.O2MAssigner(), sda.TaskAlignedFocalLoss()
    x = logits.view(-1, C).clone().requires_grad_(True)

    def assign():
        return asg.assign_batch(bp, logits.sigmoid(), gts, labs, metas)

    def loss(t):
        x.grad = None
        crit.forward_logits(x, t["labels_full"].view(-1), t["norm_metrics"].view(-1), avg_factor=10.0).backward()

    t = assign()
    loss(t)
    res = {}
    for name, fn in (("o2m_assign_batch_us", assign), ("tal_loss_fwd_bwd_us", lambda: loss(t))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / iters
    res["tal_loss_alg_gbs"] = 2 * 4 * x.numel() / (res["tal_loss_fwd_bwd_us"] * 1e-6) / 1e9
    res["positives"] = int((t["gt_inds"] > 0).sum())
    res["what"] = "35 problems x (900 queries, 80 classes, G~U[1,15]); event-timed incl. host-side packing"
    return res


def kernel_symbols(group):
    """Device kernels behind an event-timed group (names as rocprofv3 prints them, profiles/r01_bench_kernel_stats.txt)."""
    if group.startswith("msda_fwd_enc"):
        return ["msda_fwd_d32<1, 4, 408, LocAttnIO>"]
    if group.startswith("msda_bwd_enc"):
        return ["__amd_rocclr_fillBufferAligned", "msda_bwd_gather_d32<LocAttnIO, 16, 408>",
                "msda_bwd_scatter_d32_win<LocAttnIO, 16, 16, 32, 32>"]
    if group.startswith("msda_fwd_dec"):
        return ["msda_fwd_d32<1|2|4, 4, 0, LocAttnIO>"]
    if group.startswith("msda_bwd_dec"):
        return ["__amd_rocclr_fillBufferAligned", "msda_bwd_d32<8|32, LocAttnIO>"]
    return []


def cpu_baseline():
    """The oracle (a C port of the reference arithmetic; the reference itself has no native CPU path --
    ms_deform_attn_cpu.cpp:26,39 only raises) timed on this box's host cores with OpenMP on a bounded sample:
    one image of the encoder shape and one of the decoder shape, forward + backward (backward scatter by
    `omp atomic`).  Extrapolated linearly in batch to the launches of one step -> images/s."""
    import oracle
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    shapes = np.asarray(LEVELS, np.int64)
    value = (rng.random((1, S, M, D)) * 0.01).astype(np.float32)
    t = {}
    for kind, lq in (("enc", S), ("dec", NUM_QUERY + DN_PAD)):
        loc = rng.random((1, lq, M, L, P, 2)).astype(np.float32)
        a = rng.random((1, lq, M, L, P)).astype(np.float32)
        a /= a.sum((-1, -2), keepdims=True)
        go = rng.random((1, lq, M * D)).astype(np.float32)
        oracle.msda_forward(value, shapes, loc, a)                      # warm the thread pool / page in
        reps = 3 if kind == "enc" else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            oracle.msda_forward(value, shapes, loc, a)
        t[kind + "_f"] = (time.perf_counter() - t0) / reps
        t0 = time.perf_counter()
        for _ in range(reps):
            oracle.msda_backward(value, shapes, loc, a, go, parallel=True)
        t[kind + "_b"] = (time.perf_counter() - t0) / reps
    fwd_imgs, bwd_imgs = 6 * (1 + 4 * 4), 6 * (1 + 4)          # image-layers per step (enc and dec alike)
    step_s = fwd_imgs * (t["enc_f"] + t["dec_f"]) + bwd_imgs * (t["enc_b"] + t["dec_b"])
    return {"value": IMAGES_PER_GPU / step_s, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "oracle (C + OpenMP, %d threads) msda fwd+bwd on 1 encoder-shape image (Lq=22223, 3 reps) and "
                      "1 decoder-shape image (Lq=1100, 20 reps), extrapolated to the step's 102 fwd / 30 bwd "
                      "image-layers; matcher/EMA excluded" % cores,
            "enc_fwd_s": t["enc_f"], "enc_bwd_s": t["enc_b"], "dec_fwd_s": t["dec_f"], "dec_bwd_s": t["dec_b"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-micro", action="store_true")
    args = ap.parse_args()

    from semi_detr_amd import dp
    rank, local_rank, world = dp.init_distributed()
    assert world == max(1, args.gpus) or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    if os.environ.get("SEMIDETR_BENCH_SHARE_GPU"):      # test aid: all ranks on cuda:0 (with the gloo backend)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    wl = Workload(dev, seed=1234 + rank)
    reducer = None
    if world > 1:
        grads = torch.randn(GRAD_ELEMS, device=dev)
        reducer = dp.GradAllReducer(grads, bucket_bytes=64 << 20)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step(reducer)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(reducer, record=True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * IMAGES_PER_GPU * args.steps / elapsed
        stats = wl.group_stats()
        msda = {k: v for k, v in stats.items() if k.startswith("msda_")}
        dom_name = max(msda, key=lambda k: msda[k]["ms"])
        dom = msda[dom_name]
        dur_s = dom["ms"] * 1e-3 / dom["launches"]
        achieved = dom["bytes"] / dur_s / 1e9
        traffic = None
        try:   # HBM bytes per launch from the committed PMC passes (bench.py cannot collect counters itself)
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
            traffic = pmc.get(dom_name, {}).get("hbm_bytes_corrected")
        except (OSError, ValueError):
            pass
        out = {
            "metric": "images/sec/node DINO-R50 SSOD step (hot path: MSDA fwd/bwd + Hungarian + EMA/pseudo-label)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Semi-DETR COCO-10%% teacher-student step, hot path only, per GPU 1 labeled + 4 "
                                   "unlabeled 800x1333 images: 60 MSDA fwd + 24 MSDA bwd launches (S=22223, M=8, "
                                   "D=32, L=4, P=4, Lq=22223/900/1100), 39 Hungarian problems (Q=900, G~U[1,15]), "
                                   "EMA over %d params, teacher NMS + pseudo-label filter + box warp; dense GEMMs/backbone not included"
                                   % wl.n_params,
                       "images_per_gpu": IMAGES_PER_GPU,
                       "parallelism": "dp%d image-sharded, grad all-reduce %d fp32 over RCCL" % (world, GRAD_ELEMS)
                       if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "kernel_symbols": kernel_symbols(dom_name), "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": "profiles/r01_pmc_traffic.json (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE)"
                         if traffic else None,
                         "alg_bytes_per_launch": dom["bytes"], "avg_launch_us": dur_s * 1e6,
                         "launches_timed": dom["launches"]},
            "breakdown_ms_per_step": {k: v["ms"] / args.steps for k, v in sorted(stats.items())},
            "group_gbs": {k: v["bytes"] * v["launches"] / (v["ms"] * 1e-3) / 1e9 for k, v in sorted(stats.items())
                          if v["bytes"]},
        }
        if world == 1 and not args.no_micro:      # single-GPU extras; at N > 1 the other ranks would only wait
            out["microbench"] = microbench(dev)
            out["module_fused_prologue"] = module_bench(dev)
            out["warmup_stage"] = warmup_stage_bench(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
