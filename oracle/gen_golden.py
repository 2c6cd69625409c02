#!/usr/bin/env python
"""Generate tests/golden/*.npz from the REFERENCE's own Python, imported by path.

Runs only in the build container (needs /root/reference, read-only).  The fixtures it writes are data
(inputs + expected outputs); nothing of the reference's source travels.  Re-run:

    python oracle/gen_golden.py            # writes tests/golden/*.npz

What is imported from the reference (by file path, with stub registries for the absent mmcv/mmdet):
  * detr_od/models/utils/ops/functions/ms_deform_attn_func.py  -> ms_deform_attn_core_pytorch (:41-61)
    (forward oracle; backward oracle = torch.autograd through it)
  * detr_od/models/utils/ops/modules/ms_deform_attn.py          -> MSDeformAttn (:30-126)
  * thirdparty/mmdetection/mmdet/core/bbox/match_costs/match_cost.py (FocalLossCost, BBoxL1Cost, IoUCost)
  * thirdparty/mmdetection/mmdet/core/bbox/iou_calculators/iou2d_calculator.py (bbox_overlaps)
  * thirdparty/mmdetection/mmdet/core/bbox/transforms.py (cxcywh<->xyxy)
  * thirdparty/mmdetection/mmdet/core/post_processing/bbox_nms.py (multiclass_nms) over a torch restatement of
    the un-vendored mmcv-full 1.3.16 batched_nms / nms (the surrounding arithmetic is the reference's own, the
    greedy suppression is the published algorithm: PARITY UNPINNED for the keep decisions)
  * detr_ssod/models/utils/bbox_utils.py (Transform2D.transform_bboxes)
  * detr_od/core/bbox/assigners/o2m_assigner.py (O2MAssigner) + o2m_assign_result.py
  * detr_od/models/losses/task_aligned_focal_loss.py (task_aigned_focal_loss) + mmdet losses/utils.py
  * thirdparty/mmdetection/mmdet/core/bbox/assigners/hungarian_assigner.py (HungarianAssigner.assign itself, with the real
    assign_result.py / base_assigner.py; stubs only for the registry, NiceRepr and the debug logger) -- assigned_gt_inds /
    labels in cost.npz are its outputs, incl. the no-ground-truth early-out (:108-114)
  * detr_ssod/utils/hooks/mean_teacher.py (MeanTeacher, driven through before_run / before_train_iter / after_train_iter
    with a fake runner; stubs: mmcv.parallel.is_module_wrapper, mmcv.runner.hooks.HOOKS / Hook, ..logger.log_every_n)
The assignment inside HungarianAssigner.assign comes from scipy.optimize.linear_sum_assignment (scipy 1.15.3, a third-party
dependency the reference does not vendor), exactly as hungarian_assigner.py:136 calls it.  The pseudo-label fixture (pseudo.npz) comes
from the reference's OWN `DinoDetrSSOD.extract_teacher_info` (dino_detr_ssod.py:899-939), imported with stubs for the mmdet / mmcv base
classes it does not touch and called unbound on a fake teacher (see `gen_pseudo` below) -- it is not a restatement.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _load(name, path, package=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def import_reference():
    class _Registry:
        def __init__(self, *a, **k):
            pass

        def register_module(self, *a, **k):
            return lambda c: c

    def _pkg(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    # --- op: stub the native module, then load functions + modules as a package "refops"
    sys.modules["MultiScaleDeformableAttention"] = types.ModuleType("MultiScaleDeformableAttention")
    ops = REF + "/detr_od/models/utils/ops"
    _pkg("refops")
    _pkg("refops.functions")
    func = _load("refops.functions.ms_deform_attn_func", ops + "/functions/ms_deform_attn_func.py",
                 "refops.functions")
    sys.modules["refops.functions"].MSDeformAttnFunction = func.MSDeformAttnFunction
    _pkg("refops.modules")
    modl = _load("refops.modules.ms_deform_attn", ops + "/modules/ms_deform_attn.py", "refops.modules")

    # --- matcher pieces
    bb = REF + "/thirdparty/mmdetection/mmdet/core/bbox"
    for n in ["mmdet", "mmdet.core", "mmdet.core.bbox", "mmdet.core.bbox.iou_calculators",
              "mmdet.core.bbox.match_costs"]:
        _pkg(n)
    b = types.ModuleType("mmdet.core.bbox.iou_calculators.builder")
    b.IOU_CALCULATORS = _Registry()
    sys.modules[b.__name__] = b
    iou = _load("mmdet.core.bbox.iou_calculators.iou2d_calculator",
                bb + "/iou_calculators/iou2d_calculator.py", "mmdet.core.bbox.iou_calculators")
    sys.modules["mmdet.core.bbox.iou_calculators"].bbox_overlaps = iou.bbox_overlaps
    tr = _load("mmdet.core.bbox.transforms", bb + "/transforms.py", "mmdet.core.bbox")
    b2 = types.ModuleType("mmdet.core.bbox.match_costs.builder")
    b2.MATCH_COST = _Registry()
    sys.modules[b2.__name__] = b2
    mc = _load("mmdet.core.bbox.match_costs.match_cost", bb + "/match_costs/match_cost.py",
               "mmdet.core.bbox.match_costs")
    return func, modl, mc, tr, iou


def level_start(shapes):
    hw = [h * w for h, w in shapes]
    return np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)


def msda_case(func, name, shapes, N, M, D, Lq, P, dtype, seed, loc_mode="rand", gout_mode="rand"):
    g = torch.Generator().manual_seed(seed)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = torch.rand(N, S, M, D, generator=g) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    if loc_mode == "wide":          # well outside [0,1] on both sides, exercises the zero padding
        loc = loc * 1.6 - 0.3
    elif loc_mode == "edges":       # exact 0, 1, pixel centres, pixel edges, just outside
        picks = []
        for (h, w) in shapes:
            xs = [0.0, 1.0, 0.5 / w, (w - 0.5) / w, 1.0 / w, -0.5 / w, (w + 0.5) / w, -1.0 / w, 0.5]
            ys = [0.0, 1.0, 0.5 / h, (h - 0.5) / h, 1.0 / h, -0.5 / h, (h + 0.5) / h, -1.0 / h, 0.5]
            picks.append((xs, ys))
        for l, (xs, ys) in enumerate(picks):
            ix = torch.randint(0, len(xs), (N, Lq, M, P), generator=g)
            iy = torch.randint(0, len(ys), (N, Lq, M, P), generator=g)
            loc[:, :, :, l, :, 0] = torch.tensor(xs)[ix]
            loc[:, :, :, l, :, 1] = torch.tensor(ys)[iy]
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    gout = torch.rand(N, Lq, M * D, generator=g) if gout_mode == "rand" else torch.ones(N, Lq, M * D)
    tdt = torch.float64 if dtype == "f64" else torch.float32
    value, loc, attn, gout = [t.to(tdt) for t in (value, loc, attn, gout)]
    v, lo, a = [t.clone().requires_grad_(True) for t in (value, loc, attn)]
    sh = torch.as_tensor(shapes, dtype=torch.long)
    out = func.ms_deform_attn_core_pytorch(v, sh, lo, a)
    out.backward(gout)
    return {
        f"{name}.shapes": np.asarray(shapes, np.int64), f"{name}.value": value.numpy(),
        f"{name}.loc": loc.numpy(), f"{name}.attn": attn.numpy(), f"{name}.gout": gout.numpy(),
        f"{name}.out": out.detach().numpy(), f"{name}.gvalue": v.grad.numpy(),
        f"{name}.gloc": lo.grad.numpy(), f"{name}.gattn": a.grad.numpy(),
    }


def gen_msda(func):
    d = {}
    tiny = [(6, 4), (3, 2)]
    # the reference's own test.py shapes (test.py:21-36), seed 3
    d.update(msda_case(func, "testpy_f64", tiny, 1, 2, 2, 2, 2, "f64", 3))
    d.update(msda_case(func, "testpy_f32", tiny, 1, 2, 2, 2, 2, "f32", 3))
    # one D per backward dispatch branch of the reference (test.py:85-86) kept small enough to commit
    for D in (4, 30, 32, 64, 71):
        d.update(msda_case(func, f"tiny_D{D}_f64", tiny, 1, 2, D, 2, 2, "f64", 10 + D))
    d.update(msda_case(func, "tiny_D1025_f64", [(3, 2), (1, 2)], 1, 1, 1025, 1, 2, "f64", 7))
    # zero padding / boundary behaviour
    d.update(msda_case(func, "wide_f64", [(5, 7), (3, 4), (1, 1)], 2, 3, 8, 11, 3, "f64", 21, "wide"))
    d.update(msda_case(func, "edges_f64", [(4, 6), (1, 5), (3, 1), (2, 2)], 2, 2, 4, 9, 4, "f64", 22, "edges"))
    d.update(msda_case(func, "edges_f32", [(4, 6), (1, 5), (3, 1), (2, 2)], 2, 2, 32, 9, 4, "f32", 23, "edges"))
    # five levels (COCO-full variant), D=32 fast path
    d.update(msda_case(func, "five_f32", [(7, 11), (4, 6), (3, 4), (2, 3), (1, 2)], 1, 8, 32, 13, 4, "f32", 24, "wide"))
    # down-scaled DINO shape: 8 heads x 32 ch, 4 levels x 4 points
    dino = [(9, 13), (5, 7), (3, 4), (2, 2)]
    d.update(msda_case(func, "dino_f32", dino, 2, 8, 32, 21, 4, "f32", 25))
    d.update(msda_case(func, "dino_f64", dino, 2, 8, 32, 21, 4, "f64", 25))
    d.update(msda_case(func, "dino_ones_f32", dino, 2, 8, 32, 21, 4, "f32", 26, "wide", "ones"))
    np.savez_compressed(os.path.join(OUT, "msda.npz"), **d)
    return d


FULL_LEVELS = [(100, 167), (50, 84), (25, 42), (13, 21)]


def full_inputs(seed=3):
    """Inputs of the BASELINE.json micro-benchmark shape (N=2, Lq=300, L=4, M=8, P=4, D=32, S=22223), test.py's
    distributions (test.py:33-36).  45 MB: regenerated from the seed on both sides (torch's CPU generator is
    deterministic), never committed.  tests/test_gpu_msda.py repeats exactly these calls."""
    g = torch.Generator().manual_seed(seed)
    N, M, D, Lq, L, P = 2, 8, 32, 300, 4, 4
    S = sum(h * w for h, w in FULL_LEVELS)
    value = torch.rand(N, S, M, D, generator=g) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g)
    attn = torch.rand(N, Lq, M, L, P, generator=g) + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    gout = torch.rand(N, Lq, M * D, generator=g)
    return value, loc, attn, gout


def gen_msda_full(func):
    """The reference's CPU path at the full micro-benchmark shape: outputs and the two small gradients whole,
    grad_value as every 61st row + per-(image, level) sums (the tensor itself is 45 MB)."""
    value, loc, attn, gout = full_inputs()
    v, lo, a = [t.clone().requires_grad_(True) for t in (value, loc, attn)]
    sh = torch.as_tensor(FULL_LEVELS, dtype=torch.long)
    out = func.ms_deform_attn_core_pytorch(v, sh, lo, a)
    out.backward(gout)
    gv = v.grad.reshape(2, -1, 8 * 32)
    starts = np.concatenate([[0], np.cumsum([h * w for h, w in FULL_LEVELS])])
    level_sums = np.asarray([[gv[n, starts[l]:starts[l + 1]].double().sum().item() for l in range(4)] for n in range(2)])
    np.savez_compressed(os.path.join(OUT, "msda_full.npz"), out=out.detach().numpy(), gloc=lo.grad.numpy(),
                        gattn=a.grad.numpy(), gvalue_rows=gv[:, ::61].numpy(), gvalue_level_sums=level_sums,
                        input_checksum=np.asarray([value.double().sum().item(), loc.double().sum().item(),
                                                   attn.double().sum().item(), gout.double().sum().item()]))


def gen_module(modl, func):
    """Pins MSDeformAttn.forward arithmetic around the op (modules/ms_deform_attn.py:78-126)."""
    func.MSDeformAttnFunction.apply = staticmethod(
        lambda v, sh, ls, loc, a, step: func.ms_deform_attn_core_pytorch(v, sh, loc, a))
    d = {}
    shapes = [(6, 8), (3, 4), (2, 2)]
    sh = torch.as_tensor(shapes, dtype=torch.long)
    ls = torch.as_tensor(level_start(shapes))
    S = sum(h * w for h, w in shapes)
    for name, refdim, use_mask in (("ref2", 2, False), ("ref4", 4, True)):
        torch.manual_seed(5 + refdim)
        m = modl.MSDeformAttn(d_model=32, n_levels=3, n_heads=4, n_points=2).double()
        with torch.no_grad():   # the default init zeroes these; perturb so the fixture is informative
            m.sampling_offsets.weight.normal_(0, 0.05)
            m.attention_weights.weight.normal_(0, 0.2)
            m.attention_weights.bias.normal_(0, 0.2)
        N, Lq = 2, 7
        query = torch.randn(N, Lq, 32, dtype=torch.float64)
        src = torch.randn(N, S, 32, dtype=torch.float64)
        ref = torch.rand(N, Lq, 3, refdim, dtype=torch.float64)
        if refdim == 4:
            ref[..., 2:] = ref[..., 2:] * 0.3 + 0.05
        mask = (torch.rand(N, S) < 0.2) if use_mask else None
        out = m(query, ref, src, sh, ls, mask)
        for k, v in m.state_dict().items():
            d[f"{name}.sd.{k}"] = v.numpy()
        d[f"{name}.query"], d[f"{name}.src"], d[f"{name}.ref"] = query.numpy(), src.numpy(), ref.numpy()
        d[f"{name}.shapes"] = np.asarray(shapes, np.int64)
        if mask is not None:
            d[f"{name}.mask"] = mask.numpy()
        d[f"{name}.out"] = out.detach().numpy()
    np.savez_compressed(os.path.join(OUT, "msda_module.npz"), **d)


MODULE_D32_LEVELS = [(8, 12), (4, 6), (2, 3), (1, 2)]
MODULE_D32_CASES = (("enc_ref2", 2, False, True), ("enc_ref2_mask", 2, True, True),
                    ("dec_ref4", 4, False, False), ("dec_ref4_mask", 4, True, False))


def module_d32_inputs(name, refdim, use_mask, encoder, seed=77):
    """Weights + inputs of the DINO-configuration module fixture (d_model 256, 8 heads x 32 channels, 4 levels x 4
    points, fp32), regenerated from a seed on both sides (tests/conftest.py repeats exactly these calls).  Encoder
    cases: query i IS pixel i (Lq == S, 2-d reference points = pixel centres x a valid ratio); decoder cases: 37
    queries with 4-d reference boxes."""
    g = torch.Generator().manual_seed(seed + sum(map(ord, name)))
    levels = MODULE_D32_LEVELS
    S = sum(h * w for h, w in levels)
    N, d, M, L, P = 2, 256, 8, 4, 4
    Lq = S if encoder else 37

    def rn(*s, std=1.0):
        return torch.randn(*s, generator=g) * std

    th = torch.arange(M, dtype=torch.float32) * (2.0 * np.pi / M)
    grid = torch.stack([th.cos(), th.sin()], -1)
    grid = (grid / grid.abs().max(-1, keepdim=True)[0]).view(M, 1, 1, 2).repeat(1, L, P, 1)
    grid = grid * torch.arange(1, P + 1, dtype=torch.float32).view(1, 1, -1, 1)
    sd = {"sampling_offsets.weight": rn(M * L * P * 2, d, std=0.02),
          "sampling_offsets.bias": grid.reshape(-1) + rn(M * L * P * 2, std=0.1),
          "attention_weights.weight": rn(M * L * P, d, std=0.1), "attention_weights.bias": rn(M * L * P, std=0.3),
          "value_proj.weight": rn(d, d, std=0.06), "value_proj.bias": rn(d, std=0.1),
          "output_proj.weight": rn(d, d, std=0.06), "output_proj.bias": rn(d, std=0.1)}
    query, src = rn(N, Lq, d), rn(N, S, d)
    if encoder:       # transformer.py:675-691 -- pixel centres, scaled by per-image valid ratios
        ref = torch.cat([torch.stack(torch.meshgrid((torch.arange(h) + 0.5) / h, (torch.arange(w) + 0.5) / w,
                                                    indexing="ij"), -1).flip(-1).reshape(-1, 2) for h, w in levels])
        vr = torch.rand(N, 1, L, 2, generator=g) * 0.3 + 0.7
        ref = (ref.view(1, S, 1, 2) * vr).contiguous()
    else:
        ref = torch.rand(N, Lq, L, 4, generator=g)
        ref[..., 2:] = ref[..., 2:] * 0.3 + 0.05
    mask = (torch.rand(N, S, generator=g) < 0.2) if use_mask else None
    gout = rn(N, Lq, d)
    return levels, sd, query, src, ref.float(), mask, gout


def gen_module_d32(modl, func):
    """Pins the path the product takes BY DEFAULT (MSDeformAttn.fuse_prologue: fp32, 32 channels per head) to the
    reference's own module (modules/ms_deform_attn.py:78-126) + torch.autograd: output and the gradients w.r.t.
    query, input_flatten, reference_points and all eight parameters.  Large arrays are stored as flat[::5]."""
    func.MSDeformAttnFunction.apply = staticmethod(
        lambda v, sh, ls, loc, a, step: func.ms_deform_attn_core_pytorch(v, sh, loc, a))
    d = {}
    for name, refdim, use_mask, encoder in MODULE_D32_CASES:
        levels, sd, query, src, ref, mask, gout = module_d32_inputs(name, refdim, use_mask, encoder)
        m = modl.MSDeformAttn(d_model=256, n_levels=4, n_heads=8, n_points=4)
        m.load_state_dict(sd, strict=True)
        sh = torch.as_tensor(levels, dtype=torch.long)
        ls = torch.as_tensor(level_start(levels))
        q, s_, r = [t.clone().requires_grad_(True) for t in (query, src, ref)]
        out = m(q, r, s_, sh, ls, mask)
        out.backward(gout)

        def pack(t):
            a = t.detach().numpy().reshape(-1)
            return a if a.size <= 4096 else a[::5].copy()
        d[f"{name}.out"] = out.detach().numpy()
        d[f"{name}.g_query"], d[f"{name}.g_src"], d[f"{name}.g_ref"] = pack(q.grad), pack(s_.grad), pack(r.grad)
        for k, p in m.named_parameters():
            d[f"{name}.g_{k}"] = pack(p.grad)
        d[f"{name}.input_checksum"] = np.asarray([t.double().sum().item() for t in (query, src, ref, gout)]
                                                 + [sd[k].double().sum().item() for k in sorted(sd)])
    np.savez_compressed(os.path.join(OUT, "msda_module_d32.npz"), **d)


def ref_cost(mc, tr, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_w, img_h):
    """hungarian_assigner.py:115-129 with the DINO config weights (dino_detr_r50_8x2_12e_coco.py:41-44)."""
    cls_c = mc.FocalLossCost(weight=2.0)
    reg_c = mc.BBoxL1Cost(weight=5.0, box_format="xywh")
    iou_c = mc.IoUCost(iou_mode="giou", weight=2.0)
    factor = gt_bboxes.new_tensor([img_w, img_h, img_w, img_h]).unsqueeze(0)
    c1 = cls_c(cls_pred, gt_labels)
    c2 = reg_c(bbox_pred, gt_bboxes / factor)
    c3 = iou_c(tr.bbox_cxcywh_to_xyxy(bbox_pred) * factor, gt_bboxes)
    return c1, c2, c3, c1 + c2 + c3


def load_hungarian_assigner(mc):
    """The reference's HungarianAssigner class itself (hungarian_assigner.py:15-188)."""
    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__path__ = getattr(m, "__path__", [])
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Reg:
        def register_module(self, *a, **k):
            return lambda c: c

    bb = REF + "/thirdparty/mmdetection/mmdet/core/bbox"
    mod("mmdet.core.bbox.builder", BBOX_ASSIGNERS=_Reg())
    # build_match_cost: what mmcv's build_from_cfg does for these three classes -- type -> class, rest -> kwargs
    mod("mmdet.core.bbox.match_costs",
        build_match_cost=lambda cfg: getattr(mc, cfg["type"])(**{k: v for k, v in cfg.items() if k != "type"}))
    mod("mmdet.utils")
    mod("mmdet.utils.util_mixins", NiceRepr=object)
    sys.modules["mmdet.utils"].util_mixins = sys.modules["mmdet.utils.util_mixins"]
    mod("mmdet.core.bbox.assigners")
    mod("mmdet.core.bbox.assigners.logger", log_image_with_boxes=lambda *a, **k: None)      # debug=True only
    _load("mmdet.core.bbox.assigners.assign_result", bb + "/assigners/assign_result.py", "mmdet.core.bbox.assigners")
    _load("mmdet.core.bbox.assigners.base_assigner", bb + "/assigners/base_assigner.py", "mmdet.core.bbox.assigners")
    ha = _load("mmdet.core.bbox.assigners.hungarian_assigner", bb + "/assigners/hungarian_assigner.py",
               "mmdet.core.bbox.assigners")
    # the DINO config's assigner (configs/dino_detr/dino_detr_r50_8x2_12e_coco.py:41-44)
    return ha.HungarianAssigner(cls_cost=dict(type="FocalLossCost", weight=2.0),
                                reg_cost=dict(type="BBoxL1Cost", weight=5.0, box_format="xywh"),
                                iou_cost=dict(type="IoUCost", iou_mode="giou", weight=2.0))


def gen_cost(mc, tr):
    d = {}
    ref_assigner = load_hungarian_assigner(mc)
    # the docstring known answer (match_cost.py:156-162)
    iou = mc.IoUCost()(torch.FloatTensor([[1, 1, 2, 2], [2, 2, 3, 4]]),
                       torch.FloatTensor([[0, 0, 2, 4], [1, 2, 3, 4]]))
    d["doc_ioucost"] = iou.numpy()
    cases = [(5, 3, 20), (300, 1, 80), (300, 7, 80), (900, 7, 80), (900, 30, 80), (900, 100, 80),
             (17, 40, 20), (900, 0, 80), (12, 12, 5)]
    names = []
    for k, (Q, G, C) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + k)
        img_w, img_h = (1333.0, 800.0) if k % 2 == 0 else (1201.0, 800.0)
        cxcy = torch.rand(Q, 2, generator=g)
        wh = torch.rand(Q, 2, generator=g) * 0.5 + 0.01
        bbox_pred = torch.cat([cxcy, wh], -1)
        cls_pred = torch.randn(Q, C, generator=g) * 3
        xy1 = torch.rand(G, 2, generator=g) * torch.tensor([img_w, img_h]) * 0.8
        gwh = torch.rand(G, 2, generator=g) * torch.tensor([img_w, img_h]) * 0.4 + 16
        gt = torch.cat([xy1, torch.minimum(xy1 + gwh, torch.tensor([img_w, img_h]))], -1)
        labels = torch.randint(0, C, (G,), generator=g)
        if k == 0:      # degenerate boxes: zero-area prediction and zero-area gt (eps paths, iou2d:251-259)
            bbox_pred[0, 2:] = 0
            gt[0, 2:] = gt[0, :2]
            bbox_pred[1] = torch.tensor([0.5, 0.5, 0.2, 0.2])      # identical pred pair -> tied rows
            bbox_pred[2] = bbox_pred[1]
            cls_pred[2] = cls_pred[1]
        n = f"c{k}_Q{Q}_G{G}_C{C}"
        names.append(n)
        d[f"{n}.bbox_pred"], d[f"{n}.cls_pred"] = bbox_pred.numpy(), cls_pred.numpy()
        d[f"{n}.gt_bboxes"], d[f"{n}.gt_labels"] = gt.numpy(), labels.numpy()
        d[f"{n}.img_wh"] = np.asarray([img_w, img_h], np.float32)
        # the assignment: HungarianAssigner.assign ITSELF (incl. the no-ground-truth early-out, :108-114)
        res = ref_assigner.assign(bbox_pred, cls_pred, gt, labels, dict(img_shape=(img_h, img_w, 3)))
        gi, lab = res.gt_inds.numpy().astype(np.int64), res.labels.numpy().astype(np.int64)
        if G == 0:
            rows = cols = np.zeros(0, np.int64)
        else:
            # the cost terms (same calls as assign :115-129) and scipy's row / column lists, kept for the per-term tests
            c1, c2, c3, c = ref_cost(mc, tr, bbox_pred, cls_pred, gt, labels, img_w, img_h)
            if Q * G <= 900 * 7:
                d[f"{n}.cost_cls"], d[f"{n}.cost_reg"], d[f"{n}.cost_iou"] = c1.numpy(), c2.numpy(), c3.numpy()
            d[f"{n}.cost"] = c.numpy()
            rows, cols = linear_sum_assignment(c)
            chk = np.zeros(Q, np.int64)
            chk[rows] = cols + 1
            assert np.array_equal(chk, gi), "reference assign() and its own cost + scipy disagree"
        d[f"{n}.rows"], d[f"{n}.cols"] = rows.astype(np.int64), cols.astype(np.int64)
        d[f"{n}.assigned_gt_inds"], d[f"{n}.assigned_labels"] = gi, lab
    d["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(OUT, "cost.npz"), **d)


def gen_lsap():
    """Black-box pins of scipy 1.15.3 behaviour incl. ties, inf, rectangular shapes."""
    d = {}
    mats = {
        "ones53": np.ones((5, 3)), "ones35": np.ones((3, 5)),
        "tie_a": np.array([[1, 2], [1, 2], [1, 2], [0, 5.]]),
        "tie_b": np.array([[1, 1], [1, 1], [0, 0.]]),
        "inf_ok": np.array([[np.inf, 1], [2, np.inf], [3, 4]]),
        "zeros77": np.zeros((7, 7)),
    }
    rng = np.random.default_rng(7)
    for i in range(24):
        nr, nc = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        kind = i % 4
        if kind == 0:
            c = rng.random((nr, nc))
        elif kind == 1:
            c = rng.integers(0, 3, (nr, nc)).astype(float)
        elif kind == 2:
            c = np.round(rng.random((nr, nc)) * 4) / 4
            c[rng.random((nr, nc)) < 0.08] = np.inf
        else:
            c = rng.standard_normal((nr, nc)).astype(np.float32).astype(float)
        mats[f"rnd{i}"] = c
    names = []
    for k, c in mats.items():
        try:
            r, cc = linear_sum_assignment(c)
        except ValueError:
            continue
        names.append(k)
        d[f"{k}.cost"], d[f"{k}.rows"], d[f"{k}.cols"] = c, r.astype(np.int64), cc.astype(np.int64)
    d["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(OUT, "lsap.npz"), **d)


def load_mean_teacher():
    """The reference's MeanTeacher hook class itself (detr_ssod/utils/hooks/mean_teacher.py:7-64)."""
    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__path__ = getattr(m, "__path__", [])
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Reg:
        def register_module(self, *a, **k):
            return lambda c: c

    mod("mmcv")
    mod("mmcv.parallel", is_module_wrapper=lambda m: hasattr(m, "module") and not hasattr(m, "teacher"))
    mod("mmcv.runner")
    mod("mmcv.runner.hooks", HOOKS=_Reg(), Hook=object)
    mod("refssod")
    mod("refssod.utils")
    mod("refssod.utils.logger", log_every_n=lambda *a, **k: None)
    mod("refssod.utils.hooks")
    return _load("refssod.utils.hooks.mean_teacher", REF + "/detr_ssod/utils/hooks/mean_teacher.py",
                 "refssod.utils.hooks").MeanTeacher


class _FakeRunner:
    """What MeanTeacher touches of an mmcv runner: .model, .iter, .log_buffer.output."""

    def __init__(self, model):
        self.model, self.iter = model, 0
        self.log_buffer = types.SimpleNamespace(output={})


def _ts_model(shapes, gen):
    def net():
        m = torch.nn.Module()
        for i, shp in enumerate(shapes):
            m.register_parameter(f"p{i}", torch.nn.Parameter(torch.randn(*shp, generator=gen)))
        m.p0.requires_grad_(False)                      # frozen parameters are updated too (named_parameters(), :60-64)
        m.register_buffer("buf", torch.randn(5, generator=gen))      # buffers are not
        return m
    model = torch.nn.Module()
    model.teacher, model.student = net(), net()
    return model


def gen_ema():
    """Every number comes out of the reference's own MeanTeacher: the momentum schedule from before_train_iter's log entry
    (:46-49), the update arithmetic from momentum_update (:60-64), the decay from after_train_iter (:52-58), the initial clone
    from before_run (:26-35)."""
    MT = load_mean_teacher()
    d = {}
    steps = np.asarray([0, 1, 2, 3, 4, 5, 99, 100, 998, 999, 1000, 5000], np.int64)
    g = torch.Generator().manual_seed(11)
    for wu in (0, 100):
        hook, runner = MT(momentum=0.999, interval=1, warm_up=wu), _FakeRunner(_ts_model([(2,)], g))
        sched = []
        for st in steps:
            runner.iter = int(st)
            hook.before_train_iter(runner)
            sched.append(runner.log_buffer.output["ema_momentum"])
        d[f"sched_wu{wu}"] = np.asarray(sched, np.float64)
    d["sched_steps"] = steps
    shapes = [(16, 3, 7, 7), (64,), (48, 48), (1,), (17, 5), (100, 33)]
    for mi, mom in enumerate((0.0, 0.5, 0.999, 0.9996)):
        model = _ts_model(shapes, g)
        for si in range(len(shapes)):
            d[f"m{mi}.t{si}.teacher"] = getattr(model.teacher, f"p{si}").detach().numpy().copy()
            d[f"m{mi}.t{si}.student"] = getattr(model.student, f"p{si}").detach().numpy().copy()
        MT().momentum_update(model, mom)
        for si in range(len(shapes)):
            d[f"m{mi}.t{si}.out"] = getattr(model.teacher, f"p{si}").detach().numpy().copy()
        d[f"m{mi}.momentum"] = np.float64(mom)
    # whole hook sequences: before_run, then per iteration before_train_iter -> (the optimizer moves the student) ->
    # after_train_iter; teacher snapshots + logged momenta.  The student's motion is a fixed, seeded perturbation.
    seq_cfgs = {"plain": dict(momentum=0.999, interval=1, warm_up=0),
                "warm100_int2": dict(momentum=0.9996, interval=2, warm_up=100),
                "decay": dict(momentum=0.99, interval=1, warm_up=0, decay_intervals=[3, 6], decay_factor=0.1),
                "wrapped": dict(momentum=0.999, interval=1, warm_up=2)}
    seq_shapes = [(8, 3), (5,), (4, 4, 2)]
    d["seq.names"] = np.asarray(sorted(seq_cfgs))
    d["seq.iters"] = np.int64(9)
    for name in sorted(seq_cfgs):
        cfg = seq_cfgs[name]
        gg = torch.Generator().manual_seed(500 + len(name))
        model = _ts_model(seq_shapes, gg)
        for si in range(len(seq_shapes)):
            d[f"seq.{name}.teacher0.{si}"] = getattr(model.teacher, f"p{si}").detach().numpy().copy()
            d[f"seq.{name}.student0.{si}"] = getattr(model.student, f"p{si}").detach().numpy().copy()
        d[f"seq.{name}.buf0"] = model.teacher.buf.numpy().copy()
        wrapped = types.SimpleNamespace(module=model) if name == "wrapped" else model      # is_module_wrapper path (:27-28)
        hook, runner = MT(**cfg), _FakeRunner(wrapped)
        hook.before_run(runner)
        moms, hook_moms = [], []
        for it in range(9):
            runner.iter = it
            runner.log_buffer.output.pop("ema_momentum", None)
            hook.before_train_iter(runner)
            moms.append(runner.log_buffer.output.get("ema_momentum", np.nan))      # nan: skipped by `interval`
            with torch.no_grad():
                for si in range(len(seq_shapes)):
                    getattr(model.student, f"p{si}").add_(torch.randn(*seq_shapes[si], generator=gg) * 0.1)
                    d[f"seq.{name}.student{it + 1}.{si}"] = getattr(model.student, f"p{si}").detach().numpy().copy()
            hook.after_train_iter(runner)
            hook_moms.append(hook.momentum)
            for si in range(len(seq_shapes)):
                d[f"seq.{name}.teacher{it + 1}.{si}"] = getattr(model.teacher, f"p{si}").detach().numpy().copy()
        d[f"seq.{name}.logged_momentum"] = np.asarray(moms, np.float64)
        d[f"seq.{name}.hook_momentum"] = np.asarray(hook_moms, np.float64)
        d[f"seq.{name}.buf_end"] = model.teacher.buf.numpy().copy()
        d[f"seq.{name}.cfg"] = np.asarray([cfg["momentum"], cfg["interval"], cfg["warm_up"], cfg.get("decay_factor", 0.1)]
                                          + list(cfg.get("decay_intervals") or []), np.float64)
    np.savez_compressed(os.path.join(OUT, "ema.npz"), **d)


def load_dino_detr_ssod():
    """detr_ssod/models/dino_detr_ssod.py imported as it is, with stand-in modules for what its import lines name (mmcv / mmdet /
    detr_ssod are not installed; none of those names is touched by the method called below)."""
    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    class _Registry:
        def register_module(self, *a, **k):
            return lambda cls: cls

    def _unused(*a, **k):
        raise AssertionError("stand-in called: the fixture generator must not need it")

    mod("mmcv")
    mod("mmcv.runner", get_dist_info=lambda: (0, 1))
    mod("mmcv.runner.fp16_utils", force_fp32=lambda *a, **k: (lambda f: f))
    mod("mmdet")
    mod("mmdet.core", **{n: _unused for n in ("bbox2roi", "bbox_cxcywh_to_xyxy", "bbox_xyxy_to_cxcywh", "build_assigner",
                                               "build_sampler", "multi_apply", "reduce_mean")})
    mod("mmdet.models", DETECTORS=_Registry(), build_detector=_unused)
    mod("mmdet.models.utils")
    mod("mmdet.models.utils.transformer", inverse_sigmoid=_unused)
    mod("mmdet.models.builder", build_roi_extractor=_unused)
    mod("detr_ssod")
    mod("detr_ssod.models")
    mod("detr_ssod.models.multi_stream_detector", MultiSteamDetector=type("MultiSteamDetector", (torch.nn.Module,), {}))
    mod("detr_ssod.models.utils", Transform2D=_unused, filter_invalid_class_wise=_unused, concat_all_gather=_unused)
    mod("detr_ssod.utils", log_every_n=_unused, log_image_with_boxes=_unused)
    mod("detr_ssod.utils.structure_utils", dict_split=_unused, weighted_loss=_unused)
    return _load("ref_dino_detr_ssod", os.path.join(REF, "detr_ssod", "models", "dino_detr_ssod.py"))


def gen_pseudo():
    """DinoDetrSSOD.extract_teacher_info (detr_ssod/models/dino_detr_ssod.py:893-951) ITSELF, called unbound on a stand-in `self`
    whose teacher returns the seeded proposal tables: det_bboxes / det_labels / det_scores are the reference's outputs.  `keep` =
    where those rows sit in the proposal table (rows are distinct), `thr` = the sum of the two values the reference's own
    torch.mean / torch.std calls returned (recorded through a pass-through proxy of the module's `torch`)."""
    ref = load_dino_detr_ssod()
    d = {}
    g = torch.Generator().manual_seed(13)
    names, props, labs = [], [], []
    for k, K in enumerate((300, 57, 2, 1, 0, 128)):
        xy = torch.rand(K, 2, generator=g) * 800
        wh = torch.rand(K, 2, generator=g) * 300 - (20 if k in (1, 5) else 0)   # some w/h <= 0
        score = torch.rand(K, 1, generator=g) ** 3
        box = torch.cat([xy, xy + wh, score], -1)
        if K == 0:
            box = box.new_zeros(0, 5)
        props.append(box)
        labs.append(torch.randint(0, 80, (K,), generator=g))
        names.append(f"p{k}_K{K}")

    class _TorchProxy:
        """the module's `torch`, passing everything through and remembering what mean / std returned"""
        def __init__(self):
            self.means, self.stds = [], []

        def __getattr__(self, name):
            return getattr(torch, name)

        def mean(self, *a, **k):
            r = torch.mean(*a, **k)
            self.means.append(r)
            return r

        def std(self, *a, **k):
            r = torch.std(*a, **k)
            self.stds.append(r)
            return r

    proxy = _TorchProxy()
    ref.torch = proxy
    head = types.SimpleNamespace(simple_test_bboxes=lambda feat, metas, **kw: list(zip(props, labs)))
    fake = types.SimpleNamespace(teacher=types.SimpleNamespace(extract_feat=lambda img: [torch.zeros(1, 1)], bbox_head=head), curr_step=0)
    metas = [dict(transform_matrix=np.eye(3, dtype=np.float32)) for _ in props]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # std() of one / no element
        info = ref.DinoDetrSSOD.extract_teacher_info(fake, torch.zeros(1), metas)
    ref.torch = torch
    for i, n in enumerate(names):
        box, kept = props[i], info["det_bboxes"][i]
        assert len(torch.unique(box, dim=0)) == len(box)
        keep = [int(torch.nonzero((box[:, :4] == r).all(-1) & (box[:, 4] == s)).item()) for r, s in zip(kept, info["det_scores"][i])]
        assert torch.equal(info["det_labels"][i], labs[i][keep])
        d[f"{n}.proposal"], d[f"{n}.labels"] = box.numpy(), labs[i].numpy()
        d[f"{n}.keep"] = np.asarray(keep, np.int64)
        d[f"{n}.thr"] = np.float32((proxy.means[i] + proxy.stds[i]).numpy()) if len(box) else np.float32("nan")
    d["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(OUT, "pseudo.npz"), **d)


def _stub_nms(boxes, scores, iou_threshold):
    """mmcv.ops.nms (1.3.16), offset 0, restated with torch ops: visit by descending score (stable), drop a box
    when an earlier KEPT box overlaps it by more than the threshold (devIoU: inter / (Sa + Sb - inter))."""
    order = torch.sort(scores, descending=True, stable=True)[1]
    b = boxes[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = torch.zeros(len(order), dtype=torch.bool)
    keep = []
    for i in range(len(order)):
        if sup[i]:
            continue
        keep.append(i)
        w = (torch.minimum(b[i, 2], b[i + 1:, 2]) - torch.maximum(b[i, 0], b[i + 1:, 0])).clamp(min=0)
        h = (torch.minimum(b[i, 3], b[i + 1:, 3]) - torch.maximum(b[i, 1], b[i + 1:, 1])).clamp(min=0)
        inter = w * h
        sup[i + 1:] |= inter / (area[i] + area[i + 1:] - inter) > iou_threshold
    keep = order[torch.tensor(keep, dtype=torch.long)]
    return torch.cat([boxes[keep], scores[keep, None]], -1), keep


def _stub_batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms (1.3.16) restated; split_thr = -1 (the reference's cfg, head.py:1375) selects the
    per-class branch."""
    cfg = dict(nms_cfg)
    assert cfg.pop("type", "nms") == "nms" and not cfg.pop("class_agnostic", class_agnostic)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    boxes_for_nms = boxes + offsets[:, None]
    split_thr = cfg.pop("split_thr", 10000)
    assert not boxes_for_nms.shape[0] < split_thr
    total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
    scores_after_nms = scores.new_zeros(scores.size())
    for id_ in torch.unique(idxs):
        mask = (idxs == id_).nonzero(as_tuple=False).view(-1)
        dets, keep = _stub_nms(boxes_for_nms[mask], scores[mask], cfg["iou_threshold"])
        total_mask[mask[keep]] = True
        scores_after_nms[mask[keep]] = dets[:, -1]
    keep = total_mask.nonzero(as_tuple=False).view(-1)
    scores, inds = scores_after_nms[keep].sort(descending=True, stable=True)
    keep = keep[inds]
    return torch.cat([boxes[keep], scores[:, None]], -1), keep


def gen_nms(tr):
    """_get_bboxes_single(for_pseudo_label=True) (head.py:1364-1395) restated line by line around the reference's
    own multiclass_nms; logits are multiples of 1/64 so that equal scores <=> equal logits (tie order defined)."""
    ops = types.ModuleType("mmcv.ops")
    nms_mod = types.ModuleType("mmcv.ops.nms")
    nms_mod.batched_nms = _stub_batched_nms
    for n, m in (("mmcv", types.ModuleType("mmcv")), ("mmcv.ops", ops), ("mmcv.ops.nms", nms_mod)):
        sys.modules[n] = m
    pp = _load("ref_bbox_nms", REF + "/thirdparty/mmdetection/mmdet/core/post_processing/bbox_nms.py")
    d, names = {}, []
    g = torch.Generator().manual_seed(29)
    cases = [("dino", 300, 80, (800, 1333), -5.0, 300), ("dense", 120, 20, (640, 480), 0.0, 300),
             ("few", 64, 5, (333, 500), -6.5, 300), ("none", 40, 7, (100, 100), -12.0, 300),
             ("topk", 200, 10, (512, 512), 1.0, 50), ("one", 1, 1, (64, 64), 2.0, 300)]
    for name, Q, C, (ih, iw), bias, max_per_img in cases:
        logits = torch.round((torch.randn(Q, C, generator=g) * 2.0 + bias) * 64) / 64
        cxcy = torch.rand(Q, 2, generator=g)
        k = max(Q // 6, 1)                              # clusters of near-duplicates so that NMS has work to do
        cxcy[k:] = cxcy[torch.randint(0, k, (Q - k,), generator=g)] + torch.randn(Q - k, 2, generator=g) * 0.01
        wh = torch.rand(Q, 2, generator=g) * 0.3 + 0.02
        wh[k:] = wh[torch.randint(0, k, (Q - k,), generator=g)] * (1 + torch.randn(Q - k, 2, generator=g) * 0.05)
        bbox_pred = torch.cat([cxcy, wh], -1)
        # ---- head.py:1371-1395
        cls_score = logits.sigmoid()
        nms = dict(type='nms', iou_threshold=0.6, split_thr=-1)
        score_thr = 0.01
        padding = cls_score.new_zeros(cls_score.shape[0], 1)
        cls_score = torch.cat([cls_score, padding], dim=1)
        bp = tr.bbox_cxcywh_to_xyxy(bbox_pred)
        bp[:, 0::2] = bp[:, 0::2] * iw
        bp[:, 1::2] = bp[:, 1::2] * ih
        bp[:, 0::2].clamp_(min=0, max=iw)
        bp[:, 1::2].clamp_(min=0, max=ih)
        det_bboxes, det_labels = pp.multiclass_nms(bp, cls_score, score_thr, nms, max_per_img)
        names.append(name)
        d[f"{name}.logits"], d[f"{name}.bbox_pred"] = logits.numpy(), bbox_pred.numpy()
        d[f"{name}.img_hw"] = np.asarray([ih, iw], np.float32)
        d[f"{name}.max_per_img"] = np.int64(max_per_img)
        d[f"{name}.dets"] = det_bboxes.numpy().reshape(-1, 5)
        d[f"{name}.labels"] = det_labels.numpy().astype(np.int64)
    d["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(OUT, "nms.npz"), **d)


def gen_transform():
    """Transform2D.transform_bboxes (bbox_utils.py:167-192), imported by path (BitmapMasks stubbed)."""
    for n in ["mmdet.core.mask", "mmdet.core.mask.structures"]:
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = []
            sys.modules[n] = m
    sys.modules["mmdet.core.mask.structures"].BitmapMasks = type("BitmapMasks", (), {})
    bu = _load("ref_bbox_utils", REF + "/detr_ssod/models/utils/bbox_utils.py")
    d, names = {}, []
    g = torch.Generator().manual_seed(31)
    for k, K in enumerate((37, 1, 0, 300)):
        xy = torch.rand(K, 2, generator=g) * torch.tensor([1200.0, 700.0])
        wh = torch.rand(K, 2, generator=g) * 300 + 1
        box = torch.cat([xy, xy + wh, torch.rand(K, 1, generator=g)], -1)
        # a weak->strong matrix like the pipelines produce: scale, flip, translation, small shear (+ mild projective row)
        s = 0.5 + torch.rand(1, generator=g).item()
        M = torch.tensor([[-s if k % 2 else s, 0.07 * k, 30.0 * k + (1300.0 * s if k % 2 else 0.0)],
                          [0.03 * k, s * 0.9, -12.0 * k], [0.0, 1e-5 * k, 1.0]])
        out_shape = (600 + 50 * k, 900 + 30 * k, 3)
        res = bu.Transform2D.transform_bboxes(box, M, out_shape)
        n = f"t{k}_K{K}"
        names.append(n)
        d[f"{n}.boxes"], d[f"{n}.M"] = box.numpy(), M.numpy()
        d[f"{n}.out_shape"] = np.asarray(out_shape[:2], np.float32)
        d[f"{n}.out"] = res.numpy()
    d["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(OUT, "transform.npz"), **d)


def gen_o2m(tr):
    """The reference's own O2MAssigner (detr_od/core/bbox/assigners/o2m_assigner.py), imported by path, and the
    warm-up branch of _get_target_single (dino_detr_ssod_head.py:1108-1165) restated with the same torch calls
    (PseudoSampler: pos_inds = nonzero(gt_inds > 0), pos_assigned_gt_inds = gt_inds[pos_inds] - 1)."""
    class _Reg:
        def register_module(self, *a, **k):
            return lambda c: c

    def mod(name, **attrs):
        m = sys.modules.get(name) or types.ModuleType(name)
        m.__path__ = getattr(m, "__path__", [])
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    mod("mmdet.core.bbox.builder", BBOX_ASSIGNERS=_Reg())
    mod("mmdet.core.bbox.match_costs", build_match_cost=lambda cfg: None)
    mod("mmdet.core.bbox.assigners")
    mod("mmdet.core.bbox.assigners.assign_result", AssignResult=object)
    mod("mmdet.core.bbox.assigners.base_assigner", BaseAssigner=object)
    mod("mmdet.utils")
    mod("mmdet.utils.util_mixins", NiceRepr=object)
    sys.modules["mmdet.utils"].util_mixins = sys.modules["mmdet.utils.util_mixins"]
    mod("detr_ssod")
    mod("detr_ssod.utils", log_every_n=lambda *a, **k: None, log_image_with_boxes=lambda *a, **k: None)
    mod("refo2m")
    _load("refo2m.o2m_assign_result", REF + "/detr_od/core/bbox/assigners/o2m_assign_result.py", "refo2m")
    o2m = _load("refo2m.o2m_assigner", REF + "/detr_od/core/bbox/assigners/o2m_assigner.py", "refo2m")
    assigner = o2m.O2MAssigner()
    d, names = {}, []
    g = torch.Generator().manual_seed(41)
    cases = [("dino", 900, 80, 7, (800, 1333)), ("many_gt", 300, 80, 40, (640, 480)), ("one_gt", 100, 20, 1, (333, 500)),
             ("few_q", 13, 5, 3, (200, 300)), ("no_gt", 50, 10, 0, (100, 100)), ("dup_gt", 200, 6, 6, (512, 512))]
    for name, Q, C, G, (ih, iw) in cases:
        xy = torch.rand(G, 2, generator=g) * torch.tensor([iw * 0.7, ih * 0.7])
        wh = torch.rand(G, 2, generator=g) * torch.tensor([iw * 0.25, ih * 0.25]) + 8
        gt = torch.cat([xy, xy + wh], -1)
        gl = torch.randint(0, C, (G,), generator=g)
        if name == "dup_gt":                     # overlapping ground truths: one query is a candidate of several
            gt[3:] = gt[:3] + torch.rand(3, 4, generator=g) * 6
        # predictions: most random, a third jittered copies of the ground truths so that IoUs are substantial
        bp = torch.cat([torch.rand(Q, 2, generator=g), torch.rand(Q, 2, generator=g) * 0.3 + 0.02], -1)
        if G:
            src = torch.randint(0, G, (Q // 3,), generator=g)
            f = torch.tensor([iw, ih, iw, ih], dtype=torch.float32)
            near = tr.bbox_xyxy_to_cxcywh(gt[src] / f) * (1 + torch.randn(Q // 3, 4, generator=g) * 0.08)
            bp[:Q // 3] = near.clamp(0.001, 0.999)
        prob = torch.rand(Q, C, generator=g) ** 2
        meta = dict(img_shape=(ih, iw, 3))
        res = assigner.assign(bp, prob, gt, gl, meta)
        # the teacher's two options (o2m_assigner.py:115-133): best-aligned candidate only / dynamic k
        for tag, kw in (("t1", dict(teacher_assign=True)), ("tk", dict(teacher_assign=True, multiple_pos=True))):
            rt = assigner.assign(bp, prob, gt, gl, meta, **kw)
            d[f"{name}.{tag}_gt_inds"] = rt.gt_inds.numpy().astype(np.int64)
            d[f"{name}.{tag}_labels"] = rt.labels.numpy().astype(np.int64)
            d[f"{name}.{tag}_max_overlaps"] = rt.max_overlaps.numpy()
            d[f"{name}.{tag}_assign_metrics"] = rt.assign_metrics.numpy()
        # ---- head.py:1114-1160
        INF = 100000000
        assign_ious = res.max_overlaps.clone()
        assign_ious[assign_ious == -INF] = 0
        assign_metrics = res.assign_metrics
        pos_inds = torch.nonzero(res.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
        pos_assigned_gt_inds = res.gt_inds[pos_inds] - 1
        labels = gt.new_full((Q,), C, dtype=torch.long)
        bbox_targets = torch.zeros_like(bp)
        factor = bp.new_tensor([iw, ih, iw, ih]).unsqueeze(0)
        if G:
            bbox_targets[pos_inds, :] = tr.bbox_xyxy_to_cxcywh(gt[pos_assigned_gt_inds] / factor)
            labels[pos_inds] = gl[pos_assigned_gt_inds].long()
        norm = assign_metrics.new_zeros(Q)
        for gi in torch.unique(pos_assigned_gt_inds):
            idx = pos_inds[pos_assigned_gt_inds == gi]
            pm, pi = assign_metrics[idx], assign_ious[idx]
            norm[idx] = pm / (pm.max() + 10e-8) * pi.max()
        names.append(name)
        for k, v in (("bbox_pred", bp), ("cls_prob", prob), ("gt_bboxes", gt.reshape(-1, 4)), ("gt_labels", gl),
                     ("gt_inds", res.gt_inds), ("labels", res.labels), ("max_overlaps", res.max_overlaps),
                     ("assign_metrics", res.assign_metrics), ("labels_full", labels), ("bbox_targets", bbox_targets),
                     ("norm_metrics", norm)):
            d[f"{name}.{k}"] = v.numpy()
        d[f"{name}.img_hw"] = np.asarray([ih, iw], np.float32)
    d["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(OUT, "o2m.npz"), **d)


def gen_tal():
    """The reference's own task_aigned_focal_loss (detr_od/models/losses/task_aligned_focal_loss.py:35-66) with mmdet's
    weight_reduce_loss (losses/utils.py:29-55), both imported by path; gradients by torch.autograd, once w.r.t. the
    probabilities (the module's contract) and once through the call site's sigmoid (head.py:693-694)."""
    mmcv = sys.modules.get("mmcv") or types.ModuleType("mmcv")
    mmcv.jit = lambda **k: (lambda f: f)
    sys.modules["mmcv"] = mmcv
    runner = types.ModuleType("mmcv.runner")
    runner.get_dist_info = lambda: (0, 1)
    sys.modules["mmcv.runner"] = runner

    class _Reg:
        def register_module(self, *a, **k):
            return lambda c: c
    for n in ("mmdet.models", "mmdet.models.losses"):
        if n not in sys.modules:
            m = types.ModuleType(n)
            m.__path__ = []
            sys.modules[n] = m
    b = types.ModuleType("mmdet.models.builder")
    b.LOSSES = _Reg()
    sys.modules["mmdet.models.builder"] = b
    _load("mmdet.models.losses.utils", REF + "/thirdparty/mmdetection/mmdet/models/losses/utils.py", "mmdet.models.losses")
    tal = _load("ref_tal", REF + "/detr_od/models/losses/task_aligned_focal_loss.py")
    d, names = {}, []
    g = torch.Generator().manual_seed(53)
    for name, N, C, npos in (("dino", 900, 80, 90), ("small", 37, 5, 11), ("allbg", 20, 4, 0), ("sat", 64, 3, 20)):
        logits = torch.randn(N, C, generator=g) * (6.0 if name == "sat" else 2.0) - 2.0
        labels = torch.full((N,), C, dtype=torch.long)
        pos = torch.randperm(N, generator=g)[:npos]
        labels[pos] = torch.randint(0, C, (npos,), generator=g)
        metric = torch.zeros(N)
        metric[pos] = torch.rand(npos, generator=g)
        avg = float(max(metric.sum().item(), 1.0))
        x = logits.clone().requires_grad_(True)
        prob = x.sigmoid()
        prob.retain_grad()
        loss = tal.task_aigned_focal_loss(prob, labels, metric, None, gamma=2.0, reduction="mean", avg_factor=avg)
        loss.backward()
        names.append(name)
        d[f"{name}.logits"], d[f"{name}.labels"], d[f"{name}.metric"] = logits.numpy(), labels.numpy(), metric.numpy()
        d[f"{name}.avg_factor"] = np.float64(avg)
        d[f"{name}.loss"] = np.float64(loss.item())
        d[f"{name}.grad_prob"], d[f"{name}.grad_logits"] = prob.grad.numpy(), x.grad.numpy()
    d["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(OUT, "tal_loss.npz"), **d)


def main():
    os.makedirs(OUT, exist_ok=True)
    func, modl, mc, tr, _ = import_reference()
    gen_msda(func)
    gen_msda_full(func)
    gen_module(modl, func)
    gen_module_d32(modl, func)
    gen_cost(mc, tr)
    gen_lsap()
    gen_ema()
    gen_pseudo()
    gen_nms(tr)
    gen_transform()
    gen_o2m(tr)
    gen_tal()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
