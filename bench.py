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
# BASELINE.json configs: [2]/[3] = COCO 10 % split (configs/detr_ssod/detr_ssod_dino_detr_r50_coco_120k.py:6,24: 5 images per GPU,
# sample_ratio [1, 4], four feature levels); [4] = COCO-Full (detr_ssod_dino_detr_r50_coco_full_240k.py:6,24: 8 images per GPU,
# sample_ratio [1, 1] -> 4 labeled + 4 unlabeled, FIVE feature levels: dino_detr_head.py:80,218-233 adds a second stride-2 level)
RECIPES = {
    "coco10": dict(levels=LEVELS, n_sup=1, n_unsup=4),
    "full": dict(levels=LEVELS + [(7, 11)], n_sup=4, n_unsup=4),
}
ROT = 6                                                     # distinct input sets per MSDA group: one per layer of a pass
GRAD_ELEMS = 60_000_000                                     # student DINO-R50 + projector
HBM_PEAK_GBS = 8000.0                                       # MI355X_MICROARCH.md: 8 TB/s spec
PMC_JSON = "r06_pmc_traffic.json"                           # written by tools/measure_traffic.py (rocprofv3 --pmc passes)
TA_JSON = "r06_fwd_enc_TA.json"                             # written by tools/r06_fwd_ta_evidence.sh


def msda_alg_bytes(N, Lq, backward, S=S, L=L):
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


class StudentParams(torch.nn.Module):
    """The student's parameter list (DINO-R50 + projector, ~60 M fp32) as a module FlatDDP can wrap, grouped by the
    order backward finishes them: heads, decoder, encoder, backbone."""

    def __init__(self, dev, scale=1):
        super().__init__()
        sizes = [max(1, n // scale) for n in dino_param_sizes()]      # scale > 1: the CPU test of the bucket arithmetic
        nb = 3 + 9 * 16 + 3 * 4                        # ResNet-50 entries of dino_param_sizes()
        enc = 6 * 16
        dec = 6 * 22
        split = {"backbone": sizes[:nb], "encoder": sizes[nb:nb + enc], "decoder": sizes[nb + enc:nb + enc + dec],
                 "heads": sizes[nb + enc + dec:]}
        extra = GRAD_ELEMS // scale - sum(sizes)       # projector etc.: counted with the heads
        if extra > 0:
            split["heads"] = split["heads"] + [extra]
        self.groups = {}
        for name in ("backbone", "encoder", "decoder", "heads"):          # registration order = forward order
            ps = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(n, device=dev)) for n in split[name]])
            setattr(self, "p_" + name, ps)
            self.groups[name] = list(ps)


MIXED_IMG_SHAPES = [(800, 1333), (800, 1201), (750, 1333), (704, 1066)]      # images of a padded batch; the first one fills the canvas
if os.environ.get("SEMIDETR_BENCH_NO_PADDING"):      # tools: the masked call structure on a batch in which nothing is padded (all-False masks)
    MIXED_IMG_SHAPES = [(800, 1333)] * 4


class Workload:
    def __init__(self, dev, seed, recipe="coco10", io="locattn", input_sets=ROT, masked=False):
        """masked: the reference's call structure for a batch of images of DIFFERENT sizes (SURVEY 8(d) "a variant with mixed
        img_shapes to exercise masks"): every MSDeformAttn call gets the padding mask of its batch (transformer.py:1268-1309,
        :1380 -- the reference passes one even when nothing is padded), reference points are scaled by the valid ratios
        (:675-691, :975).  io "locattn": `value.masked_fill(mask, 0)` before the op and its backward after it, as
        ops/modules/ms_deform_attn.py:95-96 does; io "raw": the mask goes into the fused kernels."""
        import semi_detr_amd as sda
        self.sda, self.dev = sda, dev
        self.masked = bool(masked)
        self.rot = ROT = max(1, int(input_sets))      # noqa: N806 (shadows the module default on purpose)
        rc = RECIPES[recipe]
        self.recipe, self.io = recipe, io
        self.levels, self.n_sup, self.n_unsup = rc["levels"], rc["n_sup"], rc["n_unsup"]
        self.images_per_gpu = self.n_sup + self.n_unsup
        self.S = sum(h * w for h, w in self.levels)
        self.L = len(self.levels)
        LV, Sx, Lx = self.levels, self.S, self.L
        g = torch.Generator(device=dev).manual_seed(seed)
        self.shapes = torch.as_tensor(LV, dtype=torch.long, device=dev)
        self.starts = torch.cat([self.shapes.new_zeros(1), (self.shapes[:, 0] * self.shapes[:, 1]).cumsum(0)[:-1]])
        # encoder: query = every pixel, samples around its own centre (sigma = 2 px of that level)
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h,
                                                    (torch.arange(w, device=dev) + 0.5) / w, indexing="ij"),
                                     -1).flip(-1).reshape(-1, 2) for h, w in LV])          # (S, 2) x,y
        inv = torch.tensor([[2.0 / w, 2.0 / h] for h, w in LV], device=dev).view(1, 1, 1, Lx, 1, 2)
        self.mask, self.valid_ratio = {}, {}

        def geometry(n):
            """padding mask (n, S) and valid ratios (n, L, 2) [w, h] of n images on the 800 x 1333 canvas (transformer.py:1268-1288,
            get_valid_ratio :1230-1237); reference points of the encoder = pixel centres / valid extent x valid ratio (:675-691)"""
            if not self.masked:
                return ref.view(1, Sx, 1, 2).expand(n, Sx, Lx, 2)
            mks, vrs, refs = [], [], []
            for h, w in LV:
                mk = torch.ones(n, h, w, dtype=torch.bool, device=dev)
                for i, (ih, iw) in enumerate(MIXED_IMG_SHAPES[:n]):
                    mk[i, :math.ceil(ih * h / 800), :math.ceil(iw * w / 1333)] = False
                mks.append(mk.flatten(1))
                vrs.append(torch.stack([(~mk[:, 0, :]).sum(1).float() / w, (~mk[:, :, 0]).sum(1).float() / h], -1))
            vr = torch.stack(vrs, 1)
            for lvl, (h, w) in enumerate(LV):
                ry, rx = torch.meshgrid((torch.arange(h, device=dev) + 0.5), (torch.arange(w, device=dev) + 0.5), indexing="ij")
                refs.append(torch.stack((rx.reshape(-1)[None] / (vr[:, None, lvl, 0] * w), ry.reshape(-1)[None] / (vr[:, None, lvl, 1] * h)), -1))
            self.mask[n] = torch.cat(mks, 1).contiguous()
            self.valid_ratio[n] = vr
            return torch.cat(refs, 1)[:, :, None] * vr[:, None]               # (n, S, L, 2)

        def rand(*s):
            return torch.rand(*s, generator=g, device=dev)

        def randn(*s):
            return torch.randn(*s, generator=g, device=dev)

        def attn(n, lq):
            a = rand(n, lq, M, Lx, P) + 1e-5
            return (a / a.sum((-1, -2), keepdim=True)).contiguous()

        # Every tensor of an MSDA group exists ROT times and the six layers of a pass take them in turn, as six layers with
        # their own activations do: a bs-1 group's 80 MB working set replayed six times would sit in the 256 MB Infinity
        # Cache.  self.t[key] is a list of ROT tensors.
        self.t = {}

        def put(key, make):
            self.t[key] = [make() for _ in range(ROT)]

        for n in sorted({1, 2, 4} if recipe == "coco10" else {self.n_sup, self.n_unsup}):
            put(("value", n), lambda: rand(n, Sx, M, D) * 0.01)
            put(("enc_gout", n), lambda: rand(n, Sx, M * D))
            eref = geometry(n)                                                 # (n, S, L, 2)
            if self.masked and io != "raw":      # what autograd saved for the op's backward: the masked copy of `value`
                self.t[("value_masked", n)] = [v.masked_fill(self.mask[n][..., None, None], 0.0) for v in self.t[("value", n)]]
            if io == "raw":      # the fused prologue's inputs: reference points, RAW offsets (pixels), RAW logits
                put(("enc_ref", n), lambda: eref.contiguous().clone())
                put(("enc_off", n), lambda: (randn(n, Sx, M, Lx, P, 2) * 2.0).contiguous())
                put(("enc_logit", n), lambda: (randn(n, Sx, M, Lx * P) * 2.0).contiguous())
            else:
                put(("enc_loc", n), lambda: (eref.reshape(n, Sx, 1, Lx, 1, 2) + randn(n, Sx, M, Lx, P, 2) * inv).contiguous())
                put(("enc_attn", n), lambda: attn(n, Sx))
            # decoder reference boxes are scaled by the valid ratios per level (transformer.py:975)
            vr4 = (torch.cat([self.valid_ratio[n]] * 2, -1).view(n, 1, Lx, 4) if self.masked
                   else torch.ones(1, 1, Lx, 4, device=dev))
            for lq in (NUM_QUERY, NUM_QUERY + DN_PAD):
                # decoder: reference boxes anywhere, offsets scaled by the box size
                def dec_set(lq=lq):
                    c = rand(n, lq, 1, 1, 1, 2)
                    wh = rand(n, lq, 1, 1, 1, 2) * 0.3 + 0.02
                    box = (torch.cat([c, wh], -1).view(n, lq, 1, 4) * vr4).expand(n, lq, Lx, 4)
                    if io == "raw":  # ref_dim 4: loc = c + off / P * wh * 0.5 (ms_deform_attn.py:106-108)
                        return (box.contiguous(),
                                (randn(n, lq, M, Lx, P, 2) * P).contiguous(), (randn(n, lq, M, Lx * P) * 2.0).contiguous())
                    bx = box.reshape(n, lq, 1, Lx, 1, 4)
                    return ((bx[..., :2] + randn(n, lq, M, Lx, P, 2) * bx[..., 2:] * 0.5).contiguous(), attn(n, lq))
                sets = [dec_set() for _ in range(ROT)]
                if io == "raw":
                    self.t[("dec_ref", n, lq)] = [x[0] for x in sets]
                    self.t[("dec_off", n, lq)] = [x[1] for x in sets]
                    self.t[("dec_logit", n, lq)] = [x[2] for x in sets]
                else:
                    self.t[("dec_loc", n, lq)] = [x[0] for x in sets]
                    self.t[("dec_attn", n, lq)] = [x[1] for x in sets]
                put(("dec_gout", n, lq), lambda: rand(n, lq, M * D))
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

        self.match_sets = [problems(self.n_unsup, 1), problems(self.n_sup, 7), problems(self.n_unsup, 7), problems(2, 7)]
        self.n_match = self.n_unsup + 7 * self.n_sup + 7 * self.n_unsup
        # teacher outputs of the unlabeled images (last decoder layer): mostly background, clustered boxes
        nu = self.n_unsup
        self.t_logits = (randn(nu, NUM_QUERY, 80) * 2.0 - 5.0).contiguous()
        k = NUM_QUERY // 8
        cxcy, wh = rand(nu, NUM_QUERY, 2), rand(nu, NUM_QUERY, 2) * 0.3 + 0.02
        src = torch.randint(0, k, (NUM_QUERY - k,), generator=g, device=dev)
        cxcy[:, k:] = cxcy[:, src] + randn(nu, NUM_QUERY - k, 2) * 0.01
        wh[:, k:] = wh[:, src] * (1 + randn(nu, NUM_QUERY - k, 2) * 0.05)
        self.t_boxes = torch.cat([cxcy, wh], -1).contiguous()
        self.t_metas = [dict(img_shape=(800, 1333, 3))] * nu
        self.warp = [torch.tensor([[-0.9, 0.0, 1200.0], [0.0, 0.9, 12.0], [0.0, 0.0, 1.0]], device=dev)] * nu
        sizes = dino_param_sizes()
        self.n_params = sum(sizes)
        self.teacher = [randn(n) for n in sizes]
        self.student = [randn(n) for n in sizes]
        self.groups = {}
        self.events = []
        self.kernels = {}          # event group -> device kernels the library reports for it

    def alg_bytes(self, n, lq, backward):
        return msda_alg_bytes(n, lq, backward, self.S, self.L)

    # -- group timing with events on the launch stream (torch's current stream is the stream we pass down)
    def _timed(self, name, launches, nbytes, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        if self.record:
            self.events.append((name, launches, nbytes, e0, e1))

    def _args(self, kind, n, lq, r):
        if kind == "enc":
            keys = [("enc_ref", n), ("enc_off", n), ("enc_logit", n)] if self.io == "raw" else [("enc_loc", n), ("enc_attn", n)]
        else:
            keys = ([("dec_ref", n, lq), ("dec_off", n, lq), ("dec_logit", n, lq)] if self.io == "raw"
                    else [("dec_loc", n, lq), ("dec_attn", n, lq)])
        return [self.t[k][r % self.rot] for k in keys]

    def _fwd(self, kind, n, lq, reps):
        import MultiScaleDeformableAttention as MSDA
        fn = MSDA.ms_deform_attn_fused_forward if self.io == "raw" else MSDA.ms_deform_attn_forward
        im2col = () if self.io == "raw" else (64,)
        calls = [(self.t[("value", n)][r % self.rot], self._args(kind, n, lq, r)) for r in range(reps)]      # layer r -> input set r

        mask = self.mask[n] if self.masked else None
        mask4 = mask[..., None, None] if self.masked else None

        def run():
            for v, a in calls:
                if not self.masked:
                    fn(v, self.shapes, self.starts, *a, *im2col)
                elif self.io == "raw":
                    fn(v, self.shapes, self.starts, *a, mask)
                else:      # ops/modules/ms_deform_attn.py:95-96
                    fn(v.masked_fill(mask4, 0.0), self.shapes, self.starts, *a, *im2col)
        name = f"msda_fwd_{kind}_bs{n}_Lq{lq}"
        self._timed(name, reps, self.alg_bytes(n, lq, False), run)
        # (every time: the encoder forward's kernel is chosen from the data of the previous launches, the last call wins)
        self.kernels[name] = self.sda._lib.lib().semidetr_msda_last_kernels().decode().split("+")

    def _bwd(self, kind, n, lq, reps):
        import MultiScaleDeformableAttention as MSDA
        gk = ("enc_gout", n) if kind == "enc" else ("dec_gout", n, lq)
        fn = MSDA.ms_deform_attn_fused_backward if self.io == "raw" else MSDA.ms_deform_attn_backward
        im2col = () if self.io == "raw" else (64,)
        vk = ("value_masked", n) if self.masked and self.io != "raw" else ("value", n)
        calls = [(self.t[vk][r % self.rot], self._args(kind, n, lq, r), self.t[gk][r % self.rot]) for r in range(reps)]
        mask = self.mask[n] if self.masked else None
        mask4 = mask[..., None, None] if self.masked else None

        def run():
            for v, a, go in calls:
                if not self.masked:
                    fn(v, self.shapes, self.starts, *a, go, *im2col)
                elif self.io == "raw":
                    fn(v, self.shapes, self.starts, *a, go, mask)
                else:      # the op on the masked copy autograd saved, then masked_fill's backward on grad_value
                    fn(v, self.shapes, self.starts, *a, go, *im2col)[0].masked_fill(mask4, 0.0)
        name = f"msda_bwd_{kind}_bs{n}_Lq{lq}"
        self._timed(name, reps, self.alg_bytes(n, lq, True), run)
        if name not in self.kernels:
            self.kernels[name] = self.sda._lib.lib().semidetr_msda_last_kernels().decode().split("+")

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

    def step(self, ddp=None, record=False, reuse_encoder=False, backbone_cycles=0):
        """`ddp`: the dp.FlatDDP wrapper of the student's parameter list (None on one GPU).  The MSDA backward launches
        of this step are not autograd nodes, so the step tells the reducer which parameter groups have their final
        gradients -- through FlatDDP.mark_ready, the same bucket bookkeeping the autograd hooks drive in training
        (tests/test_dp_gloo.py runs those hooks under real autograd): decoder + heads after the decoder backward,
        encoder after the encoder backward, backbone at the end of backward (FlatDDP.finish).

        reuse_encoder: the call-site change of INTEGRATION.md 3.3 (NOT the reference's call structure, never the headline):
        the student's no-grad forward (dino_detr_ssod.py:823) and its forward_dummy (:404) see the same features and the
        same weights, and so do the teacher's simple_test_bboxes (:904) and forward_dummy (:364, :456); with dropout 0.0
        (transformer.py:1053) the second encoder pass of each model recomputes the first one's memory -- 12 of the step's
        24 unlabeled-batch encoder forwards.
        backbone_cycles: a device-side sleep of that many clock ticks between the encoder backward and the end of
        backward, standing in for the ResNet-50 backward this step omits (N > 1: its buckets then have something to
        overlap with, which is the realistic case; 0 = the pessimistic one)."""
        self.record = record
        sda = self.sda
        ns, nu, Sx = self.n_sup, self.n_unsup, self.S

        def new_batch_masks(*ns_):
            """In training every batch brings its own padding-mask tensor (transformer.py:1268-1288), so the fused path's per-mask
            summary launch (semidetr_msda_mask_extents, cached per mask tensor by the front end) runs once per batch -- three
            times per step: labeled batch, unlabeled weak (teacher), unlabeled strong (student).  A bench that kept ONE mask
            tensor alive across steps would never pay it: fresh tensors here, inside the timed region."""
            for n_ in ns_:
                if self.masked:
                    self.mask[n_] = self.mask[n_].clone()
        new_batch_masks(ns, nu)
        self._timed("ema", 1, 12 * self.n_params, lambda: sda.ema_update_(self.teacher, self.student, 0.999))
        q, qd = NUM_QUERY, NUM_QUERY + DN_PAD
        self._fwd("enc", ns, Sx, 6); self._fwd("dec", ns, qd, 6)           # supervised student forward
        self._fwd("enc", nu, Sx, 6); self._fwd("dec", nu, q, 6)            # teacher simple_test
        new_batch_masks(nu)                                               # (the student's strong-augmented views: another batch)
        self._timed("pseudo_label", 1, 0, self._pseudo)
        if not reuse_encoder:
            self._fwd("enc", nu, Sx, 6)                                    # student no-grad forward: encoder ...
        self._fwd("dec", nu, q, 6)                                         # ... and decoder
        self._timed("pseudo_label", 0, 0, self._pseudo_finish)           # lists + weak->strong warp (compute_pseudo_label_loss)
        self._match(0)                                                   # inline matching, unsup_loss
        self._fwd("enc", nu, Sx, 6); self._fwd("dec", nu, qd, 6)           # student forward_dummy (with grad)
        if not reuse_encoder:
            self._fwd("enc", nu, Sx, 6)                                    # teacher forward_dummy: encoder ...
        self._fwd("dec", nu, qd, 6)                                        # ... and decoder
        self._match(1); self._match(2)                                   # sup + unsup loss()
        self._bwd("dec", nu, qd, 6); self._bwd("dec", ns, qd, 6)
        if ddp is not None:
            ddp.mark_ready(ddp.module.groups["heads"] + ddp.module.groups["decoder"])
        self._bwd("enc", nu, Sx, 6); self._bwd("enc", ns, Sx, 6)
        if ddp is not None:
            ddp.mark_ready(ddp.module.groups["encoder"])
        # backbone_cycles == 0 is PESSIMISTIC by construction: in training the backbone's backward runs AFTER the encoder's
        # and its buckets overlap with it; without the stand-in the backbone's 60 % of the arena becomes ready only now,
        # so none of its reduction is hidden.  With it: four slices of sleep, a quarter of the backbone's parameters
        # (last layers first, the order backward finishes them) marked ready after each.
        bb = list(reversed(ddp.module.groups["backbone"])) if ddp is not None else []
        parts = 4 if backbone_cycles > 0 else 1
        for i in range(parts):
            if backbone_cycles > 0:
                torch.cuda._sleep(int(backbone_cycles) // parts)         # the backbone's backward would run here
            if ddp is not None:
                ddp.mark_ready(bb[len(bb) * i // parts:len(bb) * (i + 1) // parts])
        if ddp is not None:
            ddp.finish()

    def sup_step(self):
        """BASELINE.json config 2: fully supervised DINO-R50, 800x1333, bs 2 on one GPU -- hot path only: 6 enc + 6 dec
        MSDA forward (decoder with de-noising padding), 7 layers x 2 images of Hungarian matching, 6 + 6 MSDA backward."""
        self.record = False
        qd = NUM_QUERY + DN_PAD
        self._fwd("enc", 2, self.S, 6); self._fwd("dec", 2, qd, 6)
        self._match(3)
        self._bwd("dec", 2, qd, 6); self._bwd("enc", 2, self.S, 6)

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
    """Best streaming rate this GPU shows, GB/s, over 1 GiB arrays: the library's own float4 copy kernel (plain and
    nontemporal, semidetr_stream_copy_f32), torch's tensor.copy_, and the 2-reads-1-write EMA kernel -- the largest is
    the denominator of `frac_hbm_measured` (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy)."""
    import ctypes
    import semi_detr_amd as sda
    # the streaming probe is a measurement aid, not part of the product: it lives in the experiments build of the library
    # (include/semidetr_hip_experiments.h), loaded here ONLY for this probe -- every timed launch of the bench goes through
    # the product library
    probe = ctypes.CDLL(os.path.join(ROOT, "semi-detr_amd", "csrc", "libsemidetr_hip_exp.so"))
    probe.semidetr_stream_copy_f32.restype = ctypes.c_int
    probe.semidetr_stream_copy_f32.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int]
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    stream = sda._lib.current_stream_ptr

    def own(nt):
        rc = probe.semidetr_stream_copy_f32(stream(), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(a.data_ptr()), a.numel(), nt)
        if rc != 0:
            raise RuntimeError(f"stream_copy failed (code {rc})")

    res = {}
    bytes_per_call = {"ema_triad_2r1w": 3 * a.numel() * 4, "own_float4_read_only": a.numel() * 4}
    for name, fn in (("own_float4_copy", lambda: own(0)), ("own_float4_copy_nt", lambda: own(1)),
                     ("torch_copy", lambda: b.copy_(a)), ("ema_triad_2r1w", lambda: sda.ema_update_flat_(b, a, 0.5)),
                     ("own_float4_read_only", lambda: own(2))):
        for _ in range(3):
            fn()
        best = 0.0
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = max(best, bytes_per_call.get(name, 2 * a.numel() * 4) * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
        res[name] = best
    # the denominator is the best rate of a kernel that READS AND WRITES (the MSDA launches do both); the read-only rate
    # is reported beside it
    return max(v for k, v in res.items() if k != "own_float4_read_only"), res


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
    """BASELINE.json metric shape: N=2, Lq=300, L=4, M=8, P=4, D=32, S=22223; test.py input distributions (seed 3), fwd / bwd
    as median [p10, p90] of HIP-graph replays.  Two residency regimes, each leg against the 8 TB/s spec and against the
    streaming rate measured on this box in this run:
      * warm: ONE input set replayed -- the 45.5 MB value map lives in the 256 MB Infinity Cache, so the forward's rate is a
        cache rate, not an HBM rate (its "fraction of HBM" exceeded 1);
      * cold: 8 distinct input / output sets (8 x 47 MB > the 256 MB Infinity Cache) rotated inside the captured graph, so every
        launch finds its value map in HBM -- THIS is the HBM measurement of the BASELINE metric.
    Plus the secondary shapes SURVEY.md section 8(d) names."""
    import MultiScaleDeformableAttention as MSDA
    torch.manual_seed(3)
    N, Lq, NSETS = 2, 300, 8
    sets = [_msda_case(dev, LEVELS, N, Lq, False)[:6] for _ in range(NSETS)]
    value, shapes, starts, loc, attn, gout = sets[0]
    # ONE (spatial_shapes, level_start_index) pair for all sets, as a model has: the front end's cached level-table check is
    # keyed on these tensors (and a table first seen inside a stream capture gets the assume-nothing kernels)
    sets = [(v, shapes, starts, lo, at, go) for v, _, _, lo, at, go in sets]
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
    # (c) cold: the captured graph walks the 8 sets round robin (16 calls per replay: every set twice, 7 others in between)
    turn = [0]

    def cold(backward):
        def call():
            v, sh, st, lo, at, go = sets[turn[0] % NSETS]
            turn[0] += 1
            if backward:
                MSDA.ms_deform_attn_backward(v, sh, st, lo, at, go, 64)
            else:
                MSDA.ms_deform_attn_forward(v, sh, st, lo, at, 64)
        return call
    cold_us = {}
    for name, bw in (("fwd", False), ("bwd", True)):
        turn[0] = 0
        cold_us[name] = _graph_time(cold(bw), per_graph=2 * NSETS)
    bf, bb = msda_alg_bytes(N, Lq, False), msda_alg_bytes(N, Lq, True)
    b = bf + bb
    res["alg_bytes"] = b
    res["alg_bytes_fwd"], res["alg_bytes_bwd"] = bf, bb
    peak, peaks = hbm_stream_peak(dev)
    res["hbm_stream_measured_gbs"] = peak
    res["hbm_stream_by_method_gbs"] = peaks

    def legs(fu, bu):
        def fr(nbytes, us):
            return {"us": us, "alg_gbs": nbytes / (us * 1e-6) / 1e9, "frac_hbm_spec": nbytes / (us * 1e-6) / (HBM_PEAK_GBS * 1e9),
                    "frac_hbm_measured": nbytes / (us * 1e-6) / (peak * 1e9)}
        return {"fwd": fr(bf, fu), "bwd": fr(bb, bu), "fwd_bwd": fr(b, fu + bu)}
    res["warm"] = legs(res["fwd_us"], res["bwd_us"])
    res["warm"]["note"] = ("one input set replayed: the value map is Infinity-Cache resident, a fraction above 1 is a cache "
                           "rate, not an HBM rate")
    res["cold"] = legs(cold_us["fwd"][0], cold_us["bwd"][0])
    res["cold"]["fwd_p10_p90_us"], res["cold"]["bwd_p10_p90_us"] = list(cold_us["fwd"][1:]), list(cold_us["bwd"][1:])
    res["cold"]["note"] = "%d input sets (%.0f MB) rotated inside the captured graph: every launch reads its value map from HBM" % (
        NSETS, NSETS * (value.numel() + loc.numel() + attn.numel() + gout.numel()) * 4 / 1e6)
    # counter bytes beside the algorithmic bytes (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, tools/measure_traffic.py)
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_JSON)))
        res["counter_bytes"] = {k: pmc[k]["hbm_bytes_corrected"] for k in pmc if "micro" in k}
    except (OSError, ValueError, KeyError):
        res["counter_bytes"] = None
    # the headline fraction of the BASELINE metric = the COLD fwd + bwd against the measured streaming peak
    res["frac_hbm_peak"] = res["cold"]["fwd_bwd"]["frac_hbm_spec"]
    res["frac_hbm_measured"] = res["cold"]["fwd_bwd"]["frac_hbm_measured"]
    del sets
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


def _encoder_inputs(dev, n, img_shapes=None):
    """Encoder-layer inputs the way the DINO transformer builds them (transformer.py:675-691, :1117-1147): flattened
    pyramid, per-level padding masks from the images' true sizes inside the 800x1333 batch canvas, reference points =
    pixel centres / valid extent, scaled by the valid ratios.  img_shapes None: every image fills the canvas (no mask)."""
    shapes = torch.as_tensor(LEVELS, dtype=torch.long, device=dev)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    src = torch.randn(n, S, 256, device=dev, requires_grad=True)
    pos = torch.randn(n, S, 256, device=dev)
    if img_shapes is None:
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h,
                                                    (torch.arange(w, device=dev) + 0.5) / w, indexing="ij"), -1)
                         .flip(-1).reshape(-1, 2) for h, w in LEVELS])
        return shapes, starts, src, pos, ref.view(1, S, 1, 2).expand(n, S, L, 2).contiguous(), None
    masks, ratios = [], []
    for h, w in LEVELS:
        mk = torch.ones(n, h, w, dtype=torch.bool, device=dev)
        for i, (ih, iw) in enumerate(img_shapes):
            mk[i, :math.ceil(ih * h / 800), :math.ceil(iw * w / 1333)] = False
        masks.append(mk)
        vh, vw = (~mk[:, :, 0]).sum(1).float() / h, (~mk[:, 0, :]).sum(1).float() / w
        ratios.append(torch.stack([vw, vh], -1))
    vr = torch.stack(ratios, 1)                                           # (n, L, 2)
    refs = []
    for lvl, (h, w) in enumerate(LEVELS):
        ry, rx = torch.meshgrid(torch.linspace(0.5, h - 0.5, h, device=dev), torch.linspace(0.5, w - 0.5, w, device=dev),
                                indexing="ij")
        ry = ry.reshape(-1)[None] / (vr[:, None, lvl, 1] * h)
        rx = rx.reshape(-1)[None] / (vr[:, None, lvl, 0] * w)
        refs.append(torch.stack((rx, ry), -1))
    ref = torch.cat(refs, 1)[:, :, None] * vr[:, None]                    # (n, S, L, 2)
    return shapes, starts, src, pos, ref.contiguous(), torch.cat([m.flatten(1) for m in masks], 1)


def module_bench(dev, iters=10):
    """The nn.Module boundary: one encoder MSDeformAttn layer (bs 4, 800x1333, d_model
    256) forward + backward, incl. the four Linear layers (hipBLASLt through torch):
      * prologue/epilogue fused into the sampling kernels vs the reference's op-by-op sequence;
      * the same with mixed img_shape (padding masks + valid-ratio scaled reference points);
      * six stacked encoder layers, forward only, timed on the host clock AND with events: the reference's per-layer
        `assert (...).sum() == Len_in` is a blocking device-to-host copy; here it is answered from a cache, so the host
        runs ahead of the device (host time << device time) instead of draining the queue twelve times."""
    from semi_detr_amd import MSDeformAttn
    torch.manual_seed(0)
    m = MSDeformAttn(256, L, M, P).to(dev)
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.01)
        m.attention_weights.weight.normal_(0, 0.05)
    n = 4
    res = {}
    mixed = [(800, 1333), (800, 1201), (750, 1333), (704, 1066)]
    for tag, shp in (("", None), ("masked_", mixed)):
        shapes, starts, src, pos, ref, mask = _encoder_inputs(dev, n, shp)
        for fused in (True, False):
            m.fuse_prologue = fused
            for _ in range(2):
                m(src + pos, ref, src, shapes, starts, mask).sum().backward()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                m(src + pos, ref, src, shapes, starts, mask).sum().backward()
            e1.record()
            torch.cuda.synchronize()
            res[tag + ("fused_ms" if fused else "op_by_op_ms")] = e0.elapsed_time(e1) / iters
    res["speedup"] = res["op_by_op_ms"] / res["fused_ms"]
    # six stacked layers, forward only: host clock vs device clock
    m.fuse_prologue = True
    shapes, starts, src, pos, ref, mask = _encoder_inputs(dev, n, mixed)
    with torch.no_grad():
        x = src.detach()
        for _ in range(2):
            for _ in range(6):
                y = m(x + pos, ref, x, shapes, starts, mask)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(iters):
            for _ in range(6):
                y = m(x + pos, ref, x, shapes, starts, mask)
        e1.record()
        host_ms = (time.perf_counter() - t0) * 1e3 / iters
        torch.cuda.synchronize()
        del y
    res["stack6_fwd_device_ms"] = e0.elapsed_time(e1) / iters
    res["stack6_fwd_host_enqueue_ms"] = host_ms
    res["what"] = ("MSDeformAttn encoder layer fwd+bwd, bs4, Lq=S=22223, d_model 256, incl. Linear layers; masked_* = mixed "
                   "img_shape with padding masks; stack6 = six layers forward, host enqueue time vs device time")
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


def cpu_baseline(wl=None):
    """The oracle (a C port of the reference arithmetic -- the reference itself has no native CPU path,
    ms_deform_attn_cpu.cpp:26,39 only raises, and its Python path cannot travel to this box) timed on the host cores
    on a bounded sample: MSDA forward + backward on one encoder-shape and one decoder-shape image (backward as
    independent (head, level) tasks without atomics, and once serially), the 39 Hungarian problems of a step (oracle
    cost matrix + LSAP) and one EMA pass over the 47 M parameters.  Extrapolated linearly in images to one step."""
    # threads pinned (the runtime reads these when liboracle.so's OpenMP runtime starts, i.e. before the first import of
    # `oracle` in this process) and every leg the MEDIAN of >= 7 repetitions: 3 unpinned repetitions gave 0.0030 s and
    # 0.0101 s for the same leg on two boxes
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    import oracle
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    levels = wl.levels if wl is not None else LEVELS
    n_sup, n_unsup = (wl.n_sup, wl.n_unsup) if wl is not None else (1, 4)
    shapes = np.asarray(levels, np.int64)
    S, L = int((shapes[:, 0] * shapes[:, 1]).sum()), len(levels)
    value = (rng.random((1, S, M, D)) * 0.01).astype(np.float32)
    t, spread = {}, {}

    def med(fn, reps):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), [float(min(ts)), float(max(ts))]

    for kind, lq in (("enc", S), ("dec", NUM_QUERY + DN_PAD)):
        loc = rng.random((1, lq, M, L, P, 2)).astype(np.float32)
        a = rng.random((1, lq, M, L, P)).astype(np.float32)
        a /= a.sum((-1, -2), keepdims=True)
        go = rng.random((1, lq, M * D)).astype(np.float32)
        for _ in range(2):
            oracle.msda_forward(value, shapes, loc, a)                  # warm the thread pool / page in
            oracle.msda_backward(value, shapes, loc, a, go, parallel=True)
        reps = 9 if kind == "enc" else 21
        t[kind + "_f"], spread[kind + "_f"] = med(lambda: oracle.msda_forward(value, shapes, loc, a), reps)
        t[kind + "_b"], spread[kind + "_b"] = med(lambda: oracle.msda_backward(value, shapes, loc, a, go, parallel=True), reps)
        if kind == "enc":
            t0 = time.perf_counter()
            oracle.msda_backward(value, shapes, loc, a, go)
            t["enc_b_serial"] = time.perf_counter() - t0
    # matcher: 39 problems (Q=900, G~U[1,15]) -- cost matrix + LSAP per problem, as the reference does on the host
    probs = []
    n_match = n_unsup + 7 * (n_sup + n_unsup)
    for _ in range(n_match):
        G = int(rng.integers(1, 16))
        bp = np.concatenate([rng.random((NUM_QUERY, 2)), rng.random((NUM_QUERY, 2)) * 0.5 + 0.01], -1).astype(np.float32)
        cp = (rng.standard_normal((NUM_QUERY, 80)) * 3).astype(np.float32)
        xy = rng.random((G, 2)) * [1000, 560]
        gt = np.concatenate([xy, xy + rng.random((G, 2)) * [300, 220] + 16], -1).astype(np.float32)
        probs.append((bp, cp, gt, rng.integers(0, 80, G).astype(np.int64)))
    t0 = time.perf_counter()
    for bp, cp, gt, gl in probs:
        oracle.hungarian_assign(bp, cp, gt, gl, 1333.0, 800.0)
    t["match"] = time.perf_counter() - t0
    n_par = sum(dino_param_sizes())
    te, st = rng.random(n_par, dtype=np.float32), rng.random(n_par, dtype=np.float32)
    t0 = time.perf_counter()
    oracle.ema_update(te, st, 0.999)
    t["ema"] = time.perf_counter() - t0
    fwd_imgs, bwd_imgs = 6 * (n_sup + 4 * n_unsup), 6 * (n_sup + n_unsup)          # image-layers per step (enc and dec alike)
    step_s = fwd_imgs * (t["enc_f"] + t["dec_f"]) + bwd_imgs * (t["enc_b"] + t["dec_b"]) + t["match"] + t["ema"]
    return {"value": (n_sup + n_unsup) / step_s, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "oracle (C + OpenMP, %d threads, OMP_PROC_BIND=spread OMP_PLACES=cores): msda fwd+bwd on 1 encoder-shape image "
                      "(Lq=S, median of 9) and 1 decoder-shape image (Lq=1100, median of 21), extrapolated to the step's %d fwd / "
                      "%d bwd image-layers; "
                      "backward = independent (head, level) tasks, no atomics (serial backward stated beside it); "
                      "+ %d Hungarian problems (cost + LSAP, 1 thread) + one EMA pass over %d parameters (1 thread)"
                      % (cores, fwd_imgs, bwd_imgs, n_match, n_par),
            "enc_fwd_s": t["enc_f"], "enc_bwd_s": t["enc_b"], "enc_bwd_serial_1thread_s": t["enc_b_serial"],
            "dec_fwd_s": t["dec_f"], "dec_bwd_s": t["dec_b"], "matcher_problems_s": t["match"], "ema_s": t["ema"],
            "min_max_s": spread}


def forward_policy_bench(dev, iters=24, nsets=4, levels=None, sigmas=(1.0, 2.0, 4.0, 6.0)):
    """Encoder self-attention forward at bs 4 (the step's dominant launch) by how far the samples reach: sigma = 1 / 2 / 4 / 6 px of
    the sampled level (the kernels are level at ~5.5 px) and the reference's own initial offset STAR (`star`, `star_x2`), rotated input sets, forward and
    backward, under the three policies of semidetr_msda_set_forward_policy.  `adaptive` is
    timed after the dispatcher has seen the data (a few synchronised launches); it must sit on the faster kernel's time."""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    LEVELS = levels or globals()["LEVELS"]      # noqa: N806 (the four-level pyramid unless the caller names another)
    L, S = len(LEVELS), sum(h * w for h, w in LEVELS)      # noqa: N806
    shapes = torch.as_tensor(LEVELS, dtype=torch.long, device=dev)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h, device=dev) + 0.5) / h, (torch.arange(w, device=dev) + 0.5) / w,
                                                indexing="ij"), -1).flip(-1).reshape(-1, 2) for h, w in LEVELS])
    n, res = 4, {}
    g = torch.Generator(device=dev).manual_seed(99)
    # The reference's INITIAL offsets are not Gaussian: MSDeformAttn._reset_parameters (ops/modules/ms_deform_attn.py:62-70) sets the
    # offset bias to a star -- head m looks along direction 2 pi m / M (normalised to |.|_inf = 1), point p sits p + 1 pixels out -- and the
    # offset weights to zero, so an untrained model samples exactly there on every level (|off|_inf = 1..4 px of the sampled level).
    # `star`: that pattern; `star_x2`: the same directions twice as far (offsets that have grown, |off|_inf = 2..8 px).
    thetas = torch.arange(M, dtype=torch.float32, device=dev) * (2.0 * math.pi / M)
    star = torch.stack([thetas.cos(), thetas.sin()], -1)
    star = (star / star.abs().max(-1, keepdim=True)[0]).view(1, 1, M, 1, 1, 2) * torch.arange(1, P + 1, device=dev, dtype=torch.float32).view(1, 1, 1, 1, P, 1)
    px = torch.tensor([[1.0 / w, 1.0 / h] for h, w in LEVELS], device=dev).view(1, 1, 1, L, 1, 2)      # one pixel of each level, normalised
    cases = [("sigma_%g_px" % sg, sg, 0.0) for sg in sigmas] + [("star", 0.0, 1.0), ("star_x2", 0.0, 2.0)]
    for name, sigma, star_k in cases:
        sets = []
        for _ in range(nsets):
            a = torch.rand(n, S, M, L, P, generator=g, device=dev) + 1e-5
            off_px = torch.randn(n, S, M, L, P, 2, generator=g, device=dev) * sigma if sigma > 0 else (star * star_k).expand(n, S, M, L, P, 2)
            sets.append((torch.rand(n, S, M, D, generator=g, device=dev) * 0.01,
                         (ref.view(1, S, 1, 1, 1, 2) + off_px * px).contiguous(),
                         (a / a.sum((-1, -2), keepdim=True)).contiguous()))
        gout = torch.rand(n, S, M * D, generator=g, device=dev)
        row = {}
        for policy in ("patch", "window", "adaptive"):
            sda._lib.set_forward_policy(policy)
            for i in range(6):                   # warm-up; synchronised, so that the adaptive dispatcher sees the counts
                v, lo, at = sets[i % nsets]
                MSDA.ms_deform_attn_forward(v, shapes, starts, lo, at, 64)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                v, lo, at = sets[i % nsets]
                MSDA.ms_deform_attn_forward(v, shapes, starts, lo, at, 64)
            e1.record()
            torch.cuda.synchronize()
            row[policy + "_us"] = e0.elapsed_time(e1) * 1e3 / iters
            row[policy + "_kernel"] = sda._lib.lib().semidetr_msda_last_kernels().decode()
            # the backward behind it: its gather follows what the slot's forward launches counted (patch gather / lane-per-sample window gather)
            for i in range(2):
                v, lo, at = sets[i % nsets]
                MSDA.ms_deform_attn_backward(v, shapes, starts, lo, at, gout, 64)
            e0.record()
            for i in range(iters // 2):
                v, lo, at = sets[i % nsets]
                MSDA.ms_deform_attn_backward(v, shapes, starts, lo, at, gout, 64)
            e1.record()
            torch.cuda.synchronize()
            row[policy + "_bwd_us"] = e0.elapsed_time(e1) * 1e3 / (iters // 2)
            row[policy + "_bwd_kernels"] = sda._lib.lib().semidetr_msda_last_kernels().decode()
        row["far_fraction"] = sda._lib.forward_policy_state()["far_fraction"]
        row["adaptive_vs_patch"] = row["adaptive_us"] / row["patch_us"]
        row["adaptive_frac_hbm_peak"] = msda_alg_bytes(n, S, False, S, L) / (row["adaptive_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        res[name] = row
        del sets
    sda._lib.set_forward_policy("adaptive")
    res["what"] = ("msda forward (and the backward behind it), encoder self-attention, bs 4, Lq = S = %d, %d levels, %d rotated input sets, %d launches per timing; "
                   "far_fraction = share of level >= 1 samples further than 4 px from their query's centre, as the kernels "
                   "count it for the dispatcher; star = the reference's initial offsets (ms_deform_attn.py:62-70), star_x2 = twice as far" % (S, L, nsets, iters))
    return res


def comm_summary(ddp, stamps):
    """exposed communication + per-bucket latency of the timed steps (FlatDDP.profile), see `collectives.note`."""
    prof = []
    for st in stamps:
        if st:
            ddp._last_stamps = st
            prof.append(ddp.comm_profile())
    if not prof:
        return {"exposed_ms_per_step": None, "bucket_ready_to_done_ms": None}
    nb = max(len(p["bucket_ready_to_done_ms"]) for p in prof)
    per_bucket = [float(np.mean([p["bucket_ready_to_done_ms"][b] for p in prof if b < len(p["bucket_ready_to_done_ms"])]))
                  for b in range(nb)]
    return {"exposed_ms_per_step": float(np.mean([p["exposed_ms"] for p in prof])), "bucket_ready_to_done_ms": per_bucket}


def run_flavour(dev, seed, recipe, io, steps=5, warmup=2, reuse_encoder=False, input_sets=ROT, masked=False):
    """A short run of another flavour of the step on the same device: ms per step, images/s and the roofline fraction of
    its dominant MSDA group (algorithmic bytes / event-timed launch / 8 TB/s)."""
    wl = Workload(dev, seed, recipe=recipe, io=io, input_sets=input_sets, masked=masked)
    for _ in range(warmup):
        wl.step(reuse_encoder=reuse_encoder)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step(record=True, reuse_encoder=reuse_encoder)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    stats = {k: v for k, v in wl.group_stats().items() if k.startswith("msda_")}
    # (the dominant KERNEL as in the headline: a group of several kernels counts with its time divided among them)
    dom = max(stats, key=lambda k: stats[k]["ms"] / max(1, len([x for x in wl.kernels.get(k, []) if "fill" not in x])))
    g = stats[dom]
    us = g["ms"] * 1e3 / g["launches"]
    enc_b = stats.get("msda_bwd_enc_bs%d_Lq%d" % (wl.n_unsup, wl.S))
    enc_b_us = enc_b["ms"] * 1e3 / enc_b["launches"] if enc_b else None
    res = {"ms_per_step": dt * 1e3, "images_per_s": wl.images_per_gpu / dt, "steps": steps, "dominant_kernel": dom,
           "dominant_avg_launch_us": us, "dominant_frac_hbm_peak": g["bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "enc_bwd_avg_launch_us": enc_b_us,
           "enc_bwd_frac_hbm_peak": enc_b["bytes"] / (enc_b_us * 1e-6) / 1e9 / HBM_PEAK_GBS if enc_b else None,
           "kernels": wl.kernels.get(dom, [])}
    del wl
    torch.cuda.empty_cache()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-micro", action="store_true")
    ap.add_argument("--recipe", default="coco10", choices=sorted(RECIPES),
                    help="coco10: BASELINE.json configs[2]/[3] (1 labeled + 4 unlabeled per GPU, 4 levels); full: configs[4] "
                         "(COCO-Full: 4 labeled + 4 unlabeled per GPU, 5 feature levels)")
    ap.add_argument("--io", default="raw", choices=["locattn", "raw"],
                    help="raw (default = what a checkout runs: the product's MSDeformAttn has fuse_prologue=True): the fused "
                         "MSDeformAttn prologue / epilogue (reference points + raw offsets + raw logits); locattn: the reference op "
                         "contract (sampling locations + softmaxed weights) with the prologue outside the timed region")
    ap.add_argument("--masked", dest="masked", action="store_true", default=True,
                    help="(default) images of different sizes on the 800 x 1333 canvas: every MSDeformAttn call gets its batch's "
                         "padding mask and valid-ratio scaled reference points, as the reference's transformer passes them "
                         "(transformer.py:1268-1309; locattn: masked_fill around the op, ms_deform_attn.py:95-96; raw: the mask goes "
                         "into the fused kernels)")
    ap.add_argument("--unmasked", dest="masked", action="store_false",
                    help="no padding mask, reference points unscaled (the headline of rounds 1-5; no reference config calls the op so)")
    ap.add_argument("--reuse-encoder", action="store_true",
                    help="NOT the reference's call structure (INTEGRATION.md 3.3): the student's and the teacher's second encoder "
                         "pass over identical inputs reuse the first one's memory -- 12 fewer unlabeled-batch encoder forwards")
    ap.add_argument("--backbone-ms", type=float, default=0.0,
                    help="N > 1: a device-side sleep of this many milliseconds between the encoder backward and the end of "
                         "backward stands in for the ResNet-50 backward the step omits, so the backbone's buckets overlap with "
                         "something (realistic); 0 = pessimistic.  The sleep is subtracted from nothing: ms_per_step includes it")
    ap.add_argument("--no-flavours", action="store_true", help="skip the short runs of the other step flavours")
    ap.add_argument("--forward-policy", default="adaptive", choices=["adaptive", "patch", "window"],
                    help="encoder self-attention forward kernel (include/semidetr_hip.h: semidetr_msda_set_forward_policy); "
                         "adaptive = the library's default, chosen from how far the samples of the previous launches reached")
    ap.add_argument("--input-sets", type=int, default=ROT,
                    help="distinct input sets per MSDA group, taken in turn by the six layers of a pass (default 6: every launch "
                         "reads its inputs from HBM; 1 = the rounds 1-3 methodology, one set replayed -- the 91 MB value map of a "
                         "bs-4 group then lives in the 256 MB Infinity Cache between launches)")
    ap.add_argument("--hold", type=float, default=0.0,
                    help="keep stepping (untimed) for this many seconds after the timed steps, so that an external GPU-busy "
                         "sampler sees the device working (the timed region of a default run is ~0.15 s)")
    args = ap.parse_args()

    from semi_detr_amd import dp
    rank, local_rank, world = dp.init_distributed()
    assert world == max(1, args.gpus) or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback for the product path)"
    if os.environ.get("SEMIDETR_BENCH_SHARE_GPU"):      # test aid: all ranks on cuda:0 (with the gloo backend)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    wl = Workload(dev, seed=1234 + rank, recipe=args.recipe, io=args.io, input_sets=args.input_sets, masked=args.masked)
    wl.sda._lib.set_forward_policy(args.forward_policy)
    ipg = wl.images_per_gpu
    ddp = None
    if world > 1:
        # the drop-in for MMDistributedDataParallel (detr_ssod/apis/train.py:88-93) around the student's parameter list
        ddp = dp.FlatDDP(StudentParams(dev), broadcast_buffers=False, find_unused_parameters=False, bucket_bytes=64 << 20)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # N > 1: count the collectives a step issues (the claim is "<= 4 bucket all-reduces + <= 2 small ones per step")
    coll = {"all_reduce": 0, "all_reduce_bytes": 0, "other": 0}
    if world > 1:
        _ar, _ag, _bc = dist.all_reduce, dist.all_gather, dist.broadcast

        def counted_all_reduce(t, *a, **k):
            coll["all_reduce"] += 1
            coll["all_reduce_bytes"] += t.numel() * t.element_size()
            return _ar(t, *a, **k)

        def counted_other(fn):
            def w(*a, **k):
                coll["other"] += 1
                return fn(*a, **k)
            return w
        dist.all_reduce, dist.all_gather, dist.broadcast = counted_all_reduce, counted_other(_ag), counted_other(_bc)

    # --backbone-ms: torch.cuda._sleep takes clock ticks; calibrate ticks per millisecond on this device
    backbone_cycles = 0
    if args.backbone_ms > 0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000000)
        torch.cuda.synchronize()
        e0.record()
        torch.cuda._sleep(20000000)
        e1.record()
        torch.cuda.synchronize()
        backbone_cycles = int(args.backbone_ms * 20000000 / e0.elapsed_time(e1))
    kw = dict(reuse_encoder=args.reuse_encoder, backbone_cycles=backbone_cycles)
    if ddp is not None:
        ddp.profile = True
    for _ in range(args.warmup):
        wl.step(ddp, **kw)
    barrier()
    coll.update(all_reduce=0, all_reduce_bytes=0, other=0)
    comm_prof = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(ddp, record=True, **kw)
        if ddp is not None:
            comm_prof.append(ddp._last_stamps)       # resolved after the timed region (reading an event synchronises)
    barrier()
    elapsed = time.perf_counter() - t0
    coll_timed = dict(coll)
    t_hold = time.perf_counter()
    while args.hold > 0 and time.perf_counter() - t_hold < args.hold:      # untimed: lets a GPU-busy sampler see the device
        wl.step(ddp, **kw)
        torch.cuda.synchronize()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        ms_per_step = elapsed * 1e3 / args.steps
        value = world * ipg * args.steps / elapsed
        stats = wl.group_stats()
        msda = {k: v for k, v in stats.items() if k.startswith("msda_")}
        # the dominant KERNEL: an event group that launches several kernels (the encoder backward = gather + scatter) counts with its
        # time divided among them, so that a pair of kernels does not pass for one; the largest group's own roofline is reported next
        # to it as `roofline_largest_group` whenever the two differ (and every group's in `roofline_groups`)
        dom_name = max(msda, key=lambda k: msda[k]["ms"] / max(1, len([x for x in wl.kernels.get(k, []) if "fill" not in x])))
        big_name = max(msda, key=lambda k: msda[k]["ms"])

        def traffic_of(group):
            """HBM-side bytes per launch of an event group, from the committed PMC passes of tools/measure_traffic.sh
            (bench.py cannot collect counters itself).  The script stores the kernel names it measured; a mismatch with
            what the library launched in THIS run means the JSON is stale -> no number rather than a wrong one."""
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_JSON)))
            except (OSError, ValueError):
                return None, None
            for key in (group, group + "_patch"):       # the encoder launches have one entry per kernel choice
                e = pmc.get(key)
                if e and sorted(e.get("kernels", [])) == sorted(wl.kernels.get(group, [])):
                    return e.get("hbm_bytes_corrected"), ("profiles/" + PMC_JSON + ":" + key +
                                                          " (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, tools/measure_traffic.py)")
            return None, None

        def roofline_of(group):
            g = msda[group]
            dur_s = g["ms"] * 1e-3 / g["launches"]
            ach = g["bytes"] / dur_s / 1e9
            traffic, src = traffic_of(group)
            return {"bound": "hbm", "kernel": group, "kernel_symbols": wl.kernels.get(group, []), "achieved": ach,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                    "traffic_source": src, "alg_bytes_per_launch": g["bytes"], "avg_launch_us": dur_s * 1e6,
                    "launches_timed": g["launches"]}

        # second view of the forward: corner rows through the L1 (vector-memory data return), 64 B/clk/CU at the
        # maximum engine clock (MI355X_MICROARCH.md: 2400 MHz, 256 CUs).  Every sample reads 4 corners x 128 B.
        fwd_name = "msda_fwd_enc_bs%d_Lq%d" % (wl.n_unsup, wl.S)
        fg = msda[fwd_name]
        corner_bytes = wl.n_unsup * wl.S * M * wl.L * P * 4 * 128
        fdur = fg["ms"] * 1e-3 / fg["launches"]
        clock_mhz = torch.cuda.get_device_properties(dev).clock_rate / 1e3 if hasattr(torch.cuda.get_device_properties(dev), "clock_rate") else 2400.0
        l1_peak = 256 * 64 * 2400e6 / 1e9
        # observed clock / TA busy fraction of this kernel: measured by tools/r06_fwd_ta_evidence.sh (rocprofv3 --pmc), read
        # from the committed summary rather than carried as constants
        try:      # (one entry per forward kernel: the one this run's encoder forward launched)
            ran = (wl.kernels.get("msda_fwd_enc_bs%d_Lq%d" % (wl.n_unsup, wl.S)) or ["msda_fwd_d32"])[0].split("<")[0]
            ta_ev = json.load(open(os.path.join(ROOT, "profiles", TA_JSON)))[ran]
            obs_mhz, ta_frac = float(ta_ev["observed_clock_mhz"]), float(ta_ev["ta_busy_frac"])
        except (OSError, ValueError, KeyError, TypeError):
            ta_ev, obs_mhz, ta_frac = None, None, None
        # Second view of the encoder forward: the on-chip data paths of the kernel that RAN.  The patch kernel pulls every
        # corner row (4 x 128 B per sample) through the vector-memory data return (64 B/clk/CU).  The region-window kernel serves the
        # coarse levels' corner rows from LDS windows (ds_read_b128: 256 B/clk/CU, MI355X_MICROARCH.md) and only level 0's through the
        # vector-memory path (plus the few samples that leave their windows, not counted here): two pipes, both far from full --
        # the kernel is a latency chain at three waves per SIMD, not bound by either (ta_busy_frac_pmc says the same).
        window_fwd = any("msda_rw_d32" in x for x in wl.kernels.get(fwd_name, []))
        l1_bytes = corner_bytes / wl.L if window_fwd else corner_bytes
        lds_bytes = corner_bytes * (wl.L - 1) / wl.L if window_fwd else 0
        lds_peak = 256 * 256 * 2400e6 / 1e9
        roofline_l1 = {"bound": "l1_return", "kernel": fwd_name, "kernel_symbols": wl.kernels.get(fwd_name, []),
                       "achieved": l1_bytes / fdur / 1e9, "peak": l1_peak, "unit": "GB/s", "frac": l1_bytes / fdur / 1e9 / l1_peak,
                       "frac_at_observed_clock": (l1_bytes / fdur / 1e9 / (256 * 64 * obs_mhz * 1e6 / 1e9)) if obs_mhz else None,
                       "lds_return": ({"achieved": lds_bytes / fdur / 1e9, "peak": lds_peak, "unit": "GB/s",
                                       "frac": lds_bytes / fdur / 1e9 / lds_peak, "bytes_per_launch": lds_bytes} if window_fwd else None),
                       "observed_clock_mhz": obs_mhz, "ta_busy_frac_pmc": ta_frac,
                       "pmc_source": ("profiles/" + TA_JSON) if ta_ev else None,
                       "note": ("region-window kernel: level-0 corner rows (1 / L of the 4 x 128 B per sample; far samples of the coarse "
                                "levels not counted) through the vector-memory return vs 256 CU x 64 B/clk x 2400 MHz, the coarse levels' "
                                "corner rows from LDS (`lds_return`) vs 256 CU x 256 B/clk x 2400 MHz -- neither pipe is the limit: "
                                "timing aids (round 5) put 45 % of the kernel into geometry / records / address arithmetic, i.e. VALU and "
                                "scalar instruction issue at three waves per SIMD (DESIGN.md 2.1b, 'an INSTRUCTION problem')"
                                if window_fwd else
                                "patch kernel: corner rows (4 x 128 B per sample) / launch time vs 256 CU x 64 B/clk x 2400 MHz; this "
                                "vector-memory INSTRUCTION path (16 cycles per 64-lane buffer_load_dwordx4) is what binds it "
                                "(DESIGN.md 2.1)") +
                               "; observed clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time and TA busy = TA_BUSY_avr / active cycles, "
                               "read from profiles/" + TA_JSON + " for the kernel that ran (rocprofv3 --pmc, probe inputs; null when absent)",
                       "l1_bytes_per_launch": l1_bytes, "device_clock_mhz_reported": clock_mhz}
        out = {
            "metric": "images/sec/node DINO-R50 SSOD step (hot path: MSDA fwd/bwd + Hungarian + EMA/pseudo-label)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("Semi-DETR %s teacher-student step, hot path only, per GPU %d labeled + %d unlabeled 800x1333 "
                                    "images: 60 MSDA fwd + 24 MSDA bwd launches (S=%d, M=8, D=32, L=%d, P=4, Lq=%d/900/1100), %d "
                                    "Hungarian problems (Q=900, G~U[1,15]), EMA over %d params, teacher NMS + pseudo-label "
                                    "filter + box warp; dense GEMMs/backbone not included")
                                   % ({"coco10": "COCO-10% (BASELINE.json configs[2]/[3])", "full": "COCO-Full (BASELINE.json configs[4])"}[args.recipe],
                                      wl.n_sup, wl.n_unsup, wl.S, wl.L, wl.S, wl.n_match, wl.n_params),
                       "recipe": args.recipe, "io": args.io, "masked": bool(args.masked), "images_per_gpu": ipg, "reuse_encoder": bool(args.reuse_encoder),
                       "backbone_ms": args.backbone_ms, "input_sets_per_group": wl.rot, "forward_policy": args.forward_policy,
                       "parallelism": "dp%d image-sharded, FlatDDP bucketed grad all-reduce of %d fp32 over %s"
                                      % (world, GRAD_ELEMS, {"nccl": "RCCL (torch backend nccl)"}.get(dist.get_backend(), dist.get_backend()))
                       if world > 1 else "single GPU"},
            "roofline": roofline_of(dom_name),
            "roofline_largest_group": roofline_of(big_name) if big_name != dom_name else None,
            "roofline_l1": roofline_l1,
            "rooflines_all_msda_groups": {k: {"frac_hbm_peak": roofline_of(k)["frac"], "avg_launch_us": roofline_of(k)["avg_launch_us"],
                                              "kernels": wl.kernels.get(k, [])} for k in sorted(msda)},
            "forward_policy": wl.sda._lib.forward_policy_state(),      # encoder forward kernel choice after the timed steps
            "breakdown_ms_per_step": {k: v["ms"] / args.steps for k, v in sorted(stats.items())},
            "group_gbs": {k: v["bytes"] * v["launches"] / (v["ms"] * 1e-3) / 1e9 for k, v in sorted(stats.items())
                          if v["bytes"]},
        }
        if world > 1:
            out["collectives"] = {
                "per_step_all_reduce": coll_timed["all_reduce"] / args.steps, "per_step_other": coll_timed["other"] / args.steps,
                "all_reduce_mb_per_step": coll_timed["all_reduce_bytes"] / args.steps / 1e6,
                "backend": dist.get_backend(), "world_size": world,
                "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None,
                "nccl_env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))},
                **comm_summary(ddp, comm_prof),
                "note": "buckets of 64 MiB over a 240 MB gradient arena = 4 all-reduces per step, issued in arena order on every "
                        "rank.  exposed_ms_per_step = how long the compute stream waited for collectives after the last gradient "
                        "was final (events on that stream); bucket_ready_to_done_ms = launch -> complete per bucket, averaged over "
                        "the timed steps.  With --backbone-ms 0 the backbone's 60 % of the arena is marked ready only after the "
                        "last MSDA backward launch, so its reduction is NOT overlapped (pessimistic: in training the backbone's "
                        "backward runs behind it); --backbone-ms X puts a device-side sleep of X ms there, the backbone's "
                        "parameters becoming ready in four slices (realistic)"}
        headline_cfg = args.recipe == "coco10" and args.io == "raw" and args.masked and not args.reuse_encoder
        if world == 1 and not args.no_micro and headline_cfg:      # single-GPU extras
            # BASELINE.json config 2: supervised DINO-R50 bs 2 (hot path only), its own timed loop
            for _ in range(2):
                wl.sup_step()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                wl.sup_step()
            torch.cuda.synchronize()
            sup_s = (time.perf_counter() - t1) / args.steps
            out["supervised_dino_bs2"] = {"images_per_s": 2 / sup_s, "ms_per_step": sup_s * 1e3,
                                          "workload": "configs/dino_detr: 12 MSDA fwd + 12 MSDA bwd launches at bs 2 "
                                                      "(Lq=22223 / 1100) + 14 Hungarian problems; no EMA / pseudo labels"}
            mb = microbench(dev)
            out["microbench"] = mb
            # the BASELINE micro-benchmark (N=2, Lq=300) where a reader of the top level finds it: COLD legs (8 rotated input
            # sets, every launch reads its value map from HBM), against the 8 TB/s spec and the streaming rate measured in this
            # run, on SURVEY 8(d)'s algorithmic bytes and on the counter bytes of profiles/<PMC_JSON> (FETCH_SIZE x2 + WRITE_SIZE)
            cb = mb.get("counter_bytes") or {}
            cf = cb.get("msda_fwd_micro_bs2_Lq300_cold", cb.get("msda_fwd_micro_bs2_Lq300"))
            cbw = cb.get("msda_bwd_micro_bs2_Lq300_cold", cb.get("msda_bwd_micro_bs2_Lq300"))
            for leg, nbytes_c in (("fwd", cf), ("bwd", cbw), ("fwd_bwd", (cf + cbw) if cf and cbw else None)):
                c = mb["cold"][leg]
                out["microbench_cold_%s_us" % leg] = c["us"]
                out["microbench_cold_%s_frac_spec" % leg] = c["frac_hbm_spec"]
                out["microbench_cold_%s_frac_measured" % leg] = c["frac_hbm_measured"]
                out["microbench_cold_%s_frac_spec_counter_bytes" % leg] = (nbytes_c / (c["us"] * 1e-6) / (HBM_PEAK_GBS * 1e9)
                                                                           if nbytes_c else None)
                out["microbench_cold_%s_frac_measured_counter_bytes" % leg] = (
                    nbytes_c / (c["us"] * 1e-6) / (mb["hbm_stream_measured_gbs"] * 1e9) if nbytes_c else None)
            out["encoder_forward_by_sample_spread"] = forward_policy_bench(dev)
            out["module_fused_prologue"] = module_bench(dev)
            out["warmup_stage"] = warmup_stage_bench(dev)
        if world == 1 and not args.no_flavours and headline_cfg:
            # the other flavours of the step, 5 steps each (the driver's record then answers "COCO-Full images/s" and
            # "reference-contract images/s" by itself); the headline above is the PRODUCT DEFAULT: fused prologue + padding mask
            del wl.t
            torch.cuda.empty_cache()
            out["flavours"] = {"locattn": run_flavour(dev, 1234, "coco10", "locattn"),
                               "raw": run_flavour(dev, 1234, "coco10", "raw"),
                               "masked": run_flavour(dev, 1234, "coco10", "locattn", masked=True),
                               "full": run_flavour(dev, 1234, "full", "locattn"),
                               "full_raw": run_flavour(dev, 1234, "full", "raw"),
                               "full_raw_masked": run_flavour(dev, 1234, "full", "raw", masked=True),
                               "reuse_encoder": run_flavour(dev, 1234, "coco10", "raw", reuse_encoder=True, masked=True),
                               "replayed_inputs": run_flavour(dev, 1234, "coco10", "locattn", input_sets=1)}
            out["flavours"]["note"] = ("HEADLINE (value / ms_per_step / roofline of this line) = raw_masked: the fused MSDeformAttn "
                                       "prologue kernels the product module runs by default, with what the reference's transformer "
                                       "really passes -- the padding mask of a batch of images of different sizes in every call (inside "
                                       "the fused kernels) and valid-ratio scaled reference points: what a checkout with unchanged "
                                       "configs runs.  locattn = the reference OP contract on unmasked inputs with the prologue (softmax, "
                                       "location arithmetic, masked_fill) outside the timed region -- the headline of rounds 1-5, a call "
                                       "nobody makes; raw = the fused kernels without a mask; masked = the padded batch through the "
                                       "reference op contract, value.masked_fill(mask, 0) before the op and its backward after it "
                                       "(ms_deform_attn.py:95-96) -- what a checkout that swaps only the native module runs; full = "
                                       "COCO-Full recipe (BASELINE.json configs[4]: 4 + 4 images, five levels), reference contract; "
                                       "full_raw / full_raw_masked = the same through the fused kernels without / with the mask; "
                                       "reuse_encoder = the call-site change of INTEGRATION.md 3.3 (12 fewer encoder forwards) on the "
                                       "headline configuration, NOT the reference's call structure; replayed_inputs = the locattn step "
                                       "with ONE input set per group replayed by all six layers (--input-sets 1), the methodology of "
                                       "rounds 1-3: comparable with BENCH_r01..r03 only")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
