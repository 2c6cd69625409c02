"""Hungarian matcher of DINO-DETR / Semi-DETR on the MI355X: cost matrix, assignment and scatter stay on
the device (``csrc/match_cost.hip``, ``csrc/lsap.hip``); one launch each for a whole batch of problems.

Mirrors, name for name and argument for argument:
  * ``HungarianAssigner``            thirdparty/mmdetection/mmdet/core/bbox/assigners/hungarian_assigner.py:16-188
  * ``FocalLossCost`` / ``BBoxL1Cost`` / ``IoUCost``   .../match_costs/match_cost.py:8-50, :53-99, :146-185
  * ``AssignResult`` (fields only)   .../assigners/assign_result.py:8-60
  * ``linear_sum_assignment``        scipy.optimize (call sites hungarian_assigner.py:136, dino_detr_ssod.py:279)
  * ``O2MAssigner`` / ``O2MAssignResult``  detr_od/core/bbox/assigners/o2m_assigner.py:17-170, o2m_assign_result.py:6-52
    (the warm-up stage's one-to-many assigner, ``csrc/o2m.hip``)
and adds ``HungarianAssigner.assign_batch`` -- all (decoder layer x image) problems of a ``loss()`` call
in two launches and at most one host sync (only to raise scipy's ``ValueError`` on NaN / infeasible input).
When mmdet/mmcv are importable the classes are registered under the reference's names (see ``registry.py``).
"""
import ctypes

import torch

from . import _lib

_COST_TYPES = {}


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _require_cuda(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("semi-detr_amd matcher: tensors must live on the GPU (no CPU fallback)")


class AssignResult:
    """Same public fields as mmdet's ``AssignResult`` (assign_result.py:44-50)."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.labels = labels
        self._extra_properties = {}

    @property
    def num_preds(self):
        return len(self.gt_inds)

    def __repr__(self):
        return (f"<AssignResult(num_gts={self.num_gts}, gt_inds.shape={tuple(self.gt_inds.shape)}, "
                f"labels.shape={None if self.labels is None else tuple(self.labels.shape)})>")


# ---------------------------------------------------------------------------------------------
# functional layer over the C ABI
# ---------------------------------------------------------------------------------------------
def _params(cls_weight=0.0, alpha=0.25, gamma=2.0, eps=1e-12, reg_weight=0.0, box_format="xywh",
            iou_weight=0.0, iou_mode="giou", pred_xyxy=False):
    if box_format not in ("xyxy", "xywh"):
        raise AssertionError("box_format must be 'xyxy' or 'xywh'")
    if iou_mode not in ("iou", "giou"):
        raise AssertionError(f"Unsupported mode {iou_mode}")     # 'iof' is not used by any DETR config
    return _lib.CostParams(float(cls_weight), float(alpha), float(gamma), float(eps), float(reg_weight),
                           int(box_format == "xywh"), float(iou_weight), int(iou_mode == "giou"),
                           int(bool(pred_xyxy)))


def _to_device_async(values, dtype, device):
    """Small host list -> device tensor without a stream synchronisation (pinned staging + async copy); a
    plain ``torch.tensor(list, device=...)`` is a blocking copy that drains the stream on every call."""
    host = torch.tensor(values, dtype=dtype)
    if device.type == "cuda":
        return host.pin_memory().to(device, non_blocking=True)
    return host.to(device)


def _offsets(counts, device):
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + int(c))
    return offs, _to_device_async(offs, torch.int32, device)


def match_cost_batch(bbox_pred, cls_pred, gt_bboxes, gt_labels, gt_counts, img_wh, params):
    """bbox_pred (B,Q,4), cls_pred (B,Q,C), gt_bboxes (sumG,4), gt_labels (sumG,), gt_counts list[int],
    img_wh (B,2) tensor -> (cost_flat, gt_offsets_dev, offs_host).  Problem b's (Q,G_b) matrix is
    ``cost_flat[Q*offs[b]:Q*offs[b+1]].view(G_b, Q).t()``."""
    _require_cuda(bbox_pred, cls_pred)
    B, Q, C = cls_pred.shape
    dev = bbox_pred.device
    offs, offs_dev = _offsets(gt_counts, dev)
    total = offs[-1]
    bbox_pred = bbox_pred.detach().to(torch.float32).contiguous()
    cls_pred = cls_pred.detach().to(torch.float32).contiguous()
    gt_bboxes = gt_bboxes.detach().to(device=dev, dtype=torch.float32).reshape(-1, 4).contiguous()
    gt_labels = gt_labels.detach().to(device=dev, dtype=torch.int64).contiguous()
    img_wh = img_wh.to(device=dev, dtype=torch.float32).contiguous()
    cost = torch.empty(Q * total, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().semidetr_match_cost_f32(
            _lib.current_stream_ptr(), _p(bbox_pred), _p(cls_pred), _p(gt_bboxes), _p(gt_labels),
            _p(offs_dev), _p(img_wh), B, Q, C, total, ctypes.byref(params), _p(cost))
    _lib.check(rc, "semidetr_match_cost_f32")
    return cost, offs_dev, offs


def lsap_batch(cost_flat, offs_dev, offs, Q, gt_labels=None, want_pairs=True, want_assign=True):
    """Solve every problem of a batch on the device.  Returns dict(rows, cols, pair_offsets (host list),
    gt_inds (B,Q), labels (B,Q), status (B,) int32 device tensor)."""
    _require_cuda(cost_flat)
    dev = cost_flat.device
    B = len(offs) - 1
    counts = [offs[i + 1] - offs[i] for i in range(B)]
    total, max_gt = offs[-1], max(counts) if counts else 0
    pair_offs = [0]
    for c in counts:
        pair_offs.append(pair_offs[-1] + min(Q, c))
    lib = _lib.lib()
    rows = torch.empty(pair_offs[-1], dtype=torch.int64, device=dev) if want_pairs else None
    cols = torch.empty(pair_offs[-1], dtype=torch.int64, device=dev) if want_pairs else None
    gt_inds = torch.empty((B, Q), dtype=torch.int64, device=dev) if want_assign else None
    labels = torch.empty((B, Q), dtype=torch.int64, device=dev) if want_assign and gt_labels is not None else None
    status = torch.empty(B, dtype=torch.int32, device=dev)
    ws_bytes = lib.semidetr_lsap_workspace_bytes(B, Q, max_gt)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    if gt_labels is not None:
        gt_labels = gt_labels.detach().to(device=dev, dtype=torch.int64).contiguous()
    with torch.cuda.device(dev):
        rc = lib.semidetr_lsap_solve(_lib.current_stream_ptr(), _p(cost_flat), _p(offs_dev), _p(gt_labels), B, Q,
                                     total, max_gt, _p(rows), _p(cols), _p(gt_inds), _p(labels), _p(status),
                                     _p(ws))
    _lib.check(rc, "semidetr_lsap_solve")
    return dict(rows=rows, cols=cols, pair_offsets=pair_offs, gt_inds=gt_inds, labels=labels, status=status)


def raise_on_status(status):
    """scipy's exceptions (one host sync).  status: (B,) int32 device tensor from ``lsap_batch``."""
    bad = status.nonzero()
    if bad.numel():
        code = int(status[bad[0, 0]])
        raise ValueError("cost matrix is infeasible" if code == 1 else "matrix contains invalid numeric entries")


def linear_sum_assignment(cost_matrix, maximize=False):
    """GPU drop-in for ``scipy.optimize.linear_sum_assignment`` on ONE (Q, G) device tensor; returns
    (row_ind, col_ind) int64 device tensors, identical to scipy's result on the same matrix."""
    if maximize:
        raise NotImplementedError("maximize=True is not used on the Semi-DETR path")
    if cost_matrix.dim() != 2:
        raise ValueError("expected a matrix (2-D array), got a %r array" % (tuple(cost_matrix.shape),))
    _require_cuda(cost_matrix)
    Q, G = cost_matrix.shape
    dev = cost_matrix.device
    if Q == 0 or G == 0:
        e = torch.zeros(0, dtype=torch.int64, device=dev)
        return e, e.clone()
    if cost_matrix.dtype != torch.float32:
        # scipy up-casts fp32 exactly; wider inputs would lose bits in our fp32 storage -> refuse loudly
        raise TypeError("semi-detr_amd linear_sum_assignment takes the fp32 cost matrix the matcher builds")
    flat = cost_matrix.detach().t().contiguous().view(-1)
    offs, offs_dev = _offsets([G], dev)
    res = lsap_batch(flat, offs_dev, offs, Q, want_assign=False)
    raise_on_status(res["status"])
    return res["rows"], res["cols"]


# ---------------------------------------------------------------------------------------------
# match costs -- same constructor kwargs / call signatures as mmdet's
# ---------------------------------------------------------------------------------------------
def _dummy_boxes(n, dev):
    b = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    b[:, 2:] = 1.0
    return b


class BBoxL1Cost:
    """match_cost.py:8-50.  ``__call__(bbox_pred cxcywh normalised (Q,4), gt_bboxes xyxy normalised (G,4))``."""

    def __init__(self, weight=1.0, box_format="xyxy"):
        self.weight = weight
        assert box_format in ["xyxy", "xywh"]
        self.box_format = box_format

    def __call__(self, bbox_pred, gt_bboxes):
        Q, G, dev = bbox_pred.size(0), gt_bboxes.size(0), bbox_pred.device
        prm = _params(reg_weight=self.weight, box_format=self.box_format)
        one = torch.ones((1, 2), dtype=torch.float32, device=dev)   # gt already normalised: factor 1
        cost, _, _ = match_cost_batch(bbox_pred[None], torch.zeros((1, Q, 1), device=dev), gt_bboxes,
                                      torch.zeros(G, dtype=torch.int64, device=dev), [G], one, prm)
        return cost.view(G, Q).t()


class FocalLossCost:
    """match_cost.py:53-99.  ``__call__(cls_pred logits (Q,C), gt_labels (G,))``."""

    def __init__(self, weight=1.0, alpha=0.25, gamma=2, eps=1e-12):
        self.weight, self.alpha, self.gamma, self.eps = weight, alpha, gamma, eps

    def __call__(self, cls_pred, gt_labels):
        Q, G, dev = cls_pred.size(0), gt_labels.size(0), cls_pred.device
        prm = _params(cls_weight=self.weight, alpha=self.alpha, gamma=self.gamma, eps=self.eps)
        one = torch.ones((1, 2), dtype=torch.float32, device=dev)
        cost, _, _ = match_cost_batch(_dummy_boxes(Q, dev)[None], cls_pred[None], _dummy_boxes(G, dev), gt_labels,
                                      [G], one, prm)
        return cost.view(G, Q).t()


class IoUCost:
    """match_cost.py:146-185.  ``__call__(bboxes xyxy pixels (Q,4), gt_bboxes xyxy pixels (G,4))``."""

    def __init__(self, iou_mode="giou", weight=1.0):
        self.weight, self.iou_mode = weight, iou_mode

    def __call__(self, bboxes, gt_bboxes):
        Q, G, dev = bboxes.size(0), gt_bboxes.size(0), bboxes.device
        prm = _params(iou_weight=self.weight, iou_mode=self.iou_mode, pred_xyxy=True)
        one = torch.ones((1, 2), dtype=torch.float32, device=dev)
        cost, _, _ = match_cost_batch(bboxes[None], torch.zeros((1, Q, 1), device=dev), gt_bboxes,
                                      torch.zeros(G, dtype=torch.int64, device=dev), [G], one, prm)
        return cost.view(G, Q).t()


_COST_TYPES.update(BBoxL1Cost=BBoxL1Cost, FocalLossCost=FocalLossCost, IoUCost=IoUCost)


def build_match_cost(cfg):
    """mmcv ``build_from_cfg`` for the three cost types the DETR configs use."""
    if not isinstance(cfg, dict):
        return cfg
    args = dict(cfg)
    typ = args.pop("type")
    if typ not in _COST_TYPES:
        raise KeyError(f"{typ} is not in the match cost registry of semi-detr_amd "
                       f"(supported: {sorted(_COST_TYPES)})")
    return _COST_TYPES[typ](**args)


class HungarianAssigner:
    """hungarian_assigner.py:16-188 on the GPU.  Same constructor kwargs, same ``assign`` signature and
    return type; ``.cls_cost/.reg_cost/.iou_cost`` stay callable (dino_detr_ssod.py:265-271 calls them)."""

    def __init__(self, cls_cost=dict(type="ClassificationCost", weight=1.0),
                 reg_cost=dict(type="BBoxL1Cost", weight=1.0),
                 iou_cost=dict(type="IoUCost", iou_mode="giou", weight=1.0), debug=False):
        self.cls_cost = build_match_cost(cls_cost)
        self.reg_cost = build_match_cost(reg_cost)
        self.iou_cost = build_match_cost(iou_cost)
        if not isinstance(self.cls_cost, FocalLossCost):
            raise TypeError("semi-detr_amd HungarianAssigner implements the DINO/Semi-DETR configuration "
                            "(FocalLossCost + BBoxL1Cost + IoUCost)")
        self.debug = debug      # the reference's image-dump branch (:149-185) is a debugging aid, not ported

    def _cost_params(self):
        c, r, i = self.cls_cost, self.reg_cost, self.iou_cost
        return _params(c.weight, c.alpha, c.gamma, c.eps, r.weight, r.box_format, i.weight, i.iou_mode)

    def assign_batch(self, bbox_preds, cls_preds, gt_bboxes_list, gt_labels_list, img_metas, check=True,
                     return_cost=False):
        """All problems at once.  bbox_preds (B,Q,4), cls_preds (B,Q,C); one gt tensor pair and one img_meta
        per problem.  Returns list[AssignResult] (and the per-problem (Q,G) cost views if asked)."""
        B, Q = bbox_preds.shape[0], bbox_preds.shape[1]
        dev = bbox_preds.device
        counts = [int(g.size(0)) for g in gt_bboxes_list]
        gt_b = torch.cat([g.reshape(-1, 4) for g in gt_bboxes_list]) if B else bbox_preds.new_zeros((0, 4))
        gt_l = torch.cat([g.reshape(-1).long() for g in gt_labels_list]) if B else bbox_preds.new_zeros(0).long()
        wh = _to_device_async([[m["img_shape"][1], m["img_shape"][0]] for m in img_metas], torch.float32, dev)
        cost, offs_dev, offs = match_cost_batch(bbox_preds, cls_preds, gt_b, gt_l, counts, wh,
                                                self._cost_params())
        res = lsap_batch(cost, offs_dev, offs, Q, gt_labels=gt_l, want_pairs=return_cost)
        if check:
            raise_on_status(res["status"])
        out = []
        for b in range(B):
            gi, lab = res["gt_inds"][b], res["labels"][b]
            if Q == 0:
                gi, lab = gi.new_full((0,), -1), lab.new_full((0,), -1)
            out.append(AssignResult(counts[b], gi, None, labels=lab))
        if return_cost:
            costs = [cost[Q * offs[b]:Q * offs[b + 1]].view(counts[b], Q).t() for b in range(B)]
            return out, costs, res
        return out

    def get_targets_batch(self, bbox_preds, cls_preds, gt_bboxes_list, gt_labels_list, img_metas, num_classes,
                          check=True):
        """Assignment + training targets of all problems, device-resident: what ``get_targets`` /
        ``_get_target_single`` (dino_detr_ssod_head.py:987-1205, Hungarian branch) produce per (layer, image),
        stacked.  Returns dict(labels (B,Q), label_weights (B,Q), bbox_targets (B,Q,4), bbox_weights (B,Q,4),
        num_pos (B,) int32, gt_inds (B,Q), status).  pos_inds / neg_inds of problem b are
        ``(gt_inds[b] > 0).nonzero()`` / ``(gt_inds[b] == 0).nonzero()`` when a caller wants index lists."""
        B, Q = bbox_preds.shape[0], bbox_preds.shape[1]
        dev = bbox_preds.device
        counts = [int(g.size(0)) for g in gt_bboxes_list]
        gt_b = torch.cat([g.reshape(-1, 4) for g in gt_bboxes_list]).to(torch.float32).contiguous()
        gt_l = torch.cat([g.reshape(-1).long() for g in gt_labels_list]).contiguous()
        wh = _to_device_async([[m["img_shape"][1], m["img_shape"][0]] for m in img_metas], torch.float32, dev)
        cost, offs_dev, offs = match_cost_batch(bbox_preds, cls_preds, gt_b, gt_l, counts, wh, self._cost_params())
        res = lsap_batch(cost, offs_dev, offs, Q, gt_labels=gt_l, want_pairs=False)
        labels = torch.empty((B, Q), dtype=torch.int64, device=dev)
        label_weights = torch.empty((B, Q), dtype=torch.float32, device=dev)
        bbox_targets = torch.empty((B, Q, 4), dtype=torch.float32, device=dev)
        bbox_weights = torch.empty((B, Q, 4), dtype=torch.float32, device=dev)
        num_pos = torch.empty(B, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().semidetr_build_targets(
                _lib.current_stream_ptr(), _p(res["gt_inds"]), _p(gt_b), _p(gt_l), _p(offs_dev), _p(wh), B, Q,
                ctypes.c_int64(int(num_classes)), _p(labels), _p(label_weights), _p(bbox_targets), _p(bbox_weights),
                _p(num_pos))
        _lib.check(rc, "semidetr_build_targets")
        if check:
            raise_on_status(res["status"])
        return dict(labels=labels, label_weights=label_weights, bbox_targets=bbox_targets,
                    bbox_weights=bbox_weights, num_pos=num_pos, gt_inds=res["gt_inds"], status=res["status"])

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta, gt_bboxes_ignore=None, eps=1e-7):
        assert gt_bboxes_ignore is None, "Only case when gt_bboxes_ignore is None is supported."
        return self.assign_batch(bbox_pred[None], cls_pred[None], [gt_bboxes], [gt_labels], [img_meta])[0]


# ---------------------------------------------------------------------------------------------
# one-to-many assigner of the warm-up stage
# ---------------------------------------------------------------------------------------------
class O2MAssignResult:
    """Same public fields as the reference's ``O2MAssignResult`` (o2m_assign_result.py:44-52)."""

    def __init__(self, num_gts, gt_inds, max_overlaps, assign_metrics, labels=None):
        self.num_gts = num_gts
        self.gt_inds = gt_inds
        self.max_overlaps = max_overlaps
        self.assign_metrics = assign_metrics
        self.labels = labels
        self._extra_properties = {}

    @property
    def num_preds(self):
        return len(self.gt_inds)

    def __repr__(self):
        return (f"<O2MAssignResult(num_gts={self.num_gts}, gt_inds.shape={tuple(self.gt_inds.shape)}, "
                f"max_overlaps.shape={tuple(self.max_overlaps.shape)})>")


class O2MAssigner:
    """``O2MAssigner(candidate_topk=13)`` -- same constructor and ``assign`` signature as the reference's
    (o2m_assigner.py:46-60).  ``assign_batch`` solves every (layer, image) problem of a ``loss()`` call in one
    launch and also returns the warm-up branch's training targets (dino_detr_ssod_head.py:1114-1160)."""

    def __init__(self, candidate_topk=13, debug=False):
        self.candidate_topk = candidate_topk
        self.debug = debug

    def assign_batch(self, bbox_preds, cls_probs, gt_bboxes_list, gt_labels_list, img_metas, alpha=1, beta=6,
                     candidate_topk=None, dynamic_k=False):
        """bbox_preds (B,Q,4) normalised cxcywh, cls_probs (B,Q,C) PROBABILITIES (``cls_score.sigmoid()``), one gt
        tensor pair and one img_meta per problem.  Returns dict(gt_inds, labels, max_overlaps, assign_metrics
        (each (B,Q)), labels_full (B,Q), bbox_targets (B,Q,4), norm_metrics (B,Q), num_gts list)."""
        _require_cuda(bbox_preds, cls_probs)
        B, Q, C = cls_probs.shape
        dev = bbox_preds.device
        k = self.candidate_topk if candidate_topk is None else candidate_topk
        counts = [int(g.size(0)) for g in gt_bboxes_list]
        assert len(counts) == len(gt_labels_list) == len(img_metas) == B
        bp = bbox_preds.detach().to(torch.float32).contiguous()
        cp = cls_probs.detach().to(torch.float32).contiguous()
        gt_b = (torch.cat([g.reshape(-1, 4) for g in gt_bboxes_list]) if B else bp.new_zeros((0, 4)))
        gt_b = gt_b.detach().to(device=dev, dtype=torch.float32).contiguous()
        gt_l = (torch.cat([g.reshape(-1).long() for g in gt_labels_list]) if B else bp.new_zeros(0).long())
        gt_l = gt_l.detach().to(device=dev).contiguous()
        offs, offs_dev = _offsets(counts, dev)
        wh = _to_device_async([[m["img_shape"][1], m["img_shape"][0]] for m in img_metas] or [[1, 1]], torch.float32, dev)
        out = dict(gt_inds=torch.empty((B, Q), dtype=torch.int64, device=dev),
                   labels=torch.empty((B, Q), dtype=torch.int64, device=dev),
                   max_overlaps=torch.empty((B, Q), dtype=torch.float32, device=dev),
                   assign_metrics=torch.empty((B, Q), dtype=torch.float32, device=dev),
                   labels_full=torch.empty((B, Q), dtype=torch.int64, device=dev),
                   bbox_targets=torch.empty((B, Q, 4), dtype=torch.float32, device=dev),
                   norm_metrics=torch.empty((B, Q), dtype=torch.float32, device=dev))
        with torch.cuda.device(dev):
            rc = _lib.lib().semidetr_o2m_assign_f32(
                _lib.current_stream_ptr(), _p(bp), _p(cp), _p(gt_b), _p(gt_l), _p(offs_dev), _p(wh), B, Q, C, offs[-1],
                max(counts) if counts else 0, int(k), int(bool(dynamic_k)), float(alpha), float(beta), _p(out["gt_inds"]),
                _p(out["labels"]),
                _p(out["max_overlaps"]), _p(out["assign_metrics"]), _p(out["labels_full"]), _p(out["bbox_targets"]),
                _p(out["norm_metrics"]))
        _lib.check(rc, "semidetr_o2m_assign_f32")
        out["num_gts"] = counts
        return out

    def assign(self, bbox_pred, cls_pred, gt_bboxes, gt_labels, img_meta, gt_bboxes_ignore=None, alpha=1, beta=6,
               teacher_assign=False, multiple_pos=False):
        assert gt_bboxes_ignore is None, "Only case when gt_bboxes_ignore is None is supported."
        # o2m_assigner.py:115-133: the teacher keeps only the best-aligned candidate (option 1), or -- multiple_pos -- the
        # first k_g of the top-k candidates with k_g estimated from the ground truth's top-k IoUs (option 2)
        k = 1 if (teacher_assign and not multiple_pos) else self.candidate_topk
        r = self.assign_batch(bbox_pred[None], cls_pred[None], [gt_bboxes], [gt_labels], [img_meta], alpha, beta, k,
                              dynamic_k=bool(teacher_assign and multiple_pos))
        return O2MAssignResult(r["num_gts"][0], r["gt_inds"][0], r["max_overlaps"][0], r["assign_metrics"][0],
                               labels=r["labels"][0])
