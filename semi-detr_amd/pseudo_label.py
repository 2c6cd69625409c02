"""Teacher pseudo-label step on the MI355X (``csrc/pseudo_label.hip``, ``csrc/nms.hip``).

* ``filter_pseudo_labels`` replaces the per-image Python loop of ``DinoDetrSSOD.extract_teacher_info``
  (detr_ssod/models/dino_detr_ssod.py:918-939): threshold = mean + unbiased std of the image's scores, keep
  ``score >= thr``, drop boxes with non-positive width/height, keep post-NMS order.  One launch and one
  host read-back of the kept counts for the whole batch (the reference syncs several times per image).
* ``get_bboxes_for_pseudo_label`` replaces ``DINODETRSSODHead.get_bboxes(..., for_pseudo_label=True)`` /
  ``_get_bboxes_single`` (detr_od/models/dense_heads/dino_detr_ssod_head.py:1320-1331, :1364-1395): sigmoid,
  box decoding, ``multiclass_nms`` (mmdet bbox_nms.py:8-95 over mmcv.ops.batched_nms), top ``max_per_img``.
* ``teacher_pseudo_labels`` chains both on the device: one host sync for the whole batch.
* ``transform_bboxes`` replaces ``Transform2D.transform_bboxes`` (detr_ssod/models/utils/bbox_utils.py:167-192).
"""
import ctypes

import torch

from . import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def filter_pseudo_labels(proposal_box_list, proposal_label_list, return_threshold=False):
    """proposal_box_list: list of (K_i, 5) [x1,y1,x2,y2,score]; proposal_label_list: list of (K_i,).
    Returns (det_bboxes, det_labels, det_scores) lists exactly as extract_teacher_info builds them."""
    B = len(proposal_box_list)
    if B == 0:
        return ([], [], []) + (([],) if return_threshold else ())
    dev = proposal_box_list[0].device
    if not dev.type == "cuda":
        raise RuntimeError("filter_pseudo_labels: tensors must live on the GPU (no CPU fallback)")
    counts = [int(p.size(0)) for p in proposal_box_list]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    total = offs[-1]
    prop = torch.cat([p.reshape(-1, 5) for p in proposal_box_list]).to(torch.float32).contiguous()
    lab = torch.cat([l.reshape(-1) for l in proposal_label_list]).to(torch.int64).contiguous()
    offs_dev = torch.tensor(offs, dtype=torch.int32, device=dev)
    out_boxes = torch.empty((total, 4), dtype=torch.float32, device=dev)
    out_labels = torch.empty(total, dtype=torch.int64, device=dev)
    out_scores = torch.empty(total, dtype=torch.float32, device=dev)
    out_keep = torch.empty(total, dtype=torch.int32, device=dev)
    out_count = torch.empty(B, dtype=torch.int32, device=dev)
    out_thr = torch.empty(B, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().semidetr_pseudo_label_filter_f32(
            _lib.current_stream_ptr(), _p(prop), _p(lab), _p(offs_dev), _p(None), B, _p(out_boxes), _p(out_labels),
            _p(out_scores), _p(out_keep), _p(out_count), _p(out_thr))
    _lib.check(rc, "semidetr_pseudo_label_filter_f32")
    kept = out_count.tolist()          # the one host sync: list lengths are data dependent
    det_bboxes = [out_boxes[offs[b]:offs[b] + kept[b]] for b in range(B)]
    det_labels = [out_labels[offs[b]:offs[b] + kept[b]].to(proposal_label_list[b].dtype) for b in range(B)]
    det_scores = [out_scores[offs[b]:offs[b] + kept[b]] for b in range(B)]
    if return_threshold:
        return det_bboxes, det_labels, det_scores, out_thr
    return det_bboxes, det_labels, det_scores


def _img_hw(img_metas, dev):
    return _lib_small([[float(m["img_shape"][0]), float(m["img_shape"][1])] for m in img_metas], torch.float32, dev)


def _lib_small(values, dtype, dev):
    """Small host list -> device tensor through pinned memory (no stream sync)."""
    t = torch.tensor(values, dtype=dtype)
    return t.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else t.to(dev)


def _nms_batch(cls_scores, bbox_preds, img_metas, score_thr, iou_threshold, max_per_img):
    if cls_scores.device.type != "cuda":
        raise RuntimeError("get_bboxes_for_pseudo_label: tensors must live on the GPU (no CPU fallback)")
    if cls_scores.dim() != 3 or bbox_preds.shape != cls_scores.shape[:2] + (4,):
        raise ValueError(f"expected cls_scores (B,Q,C) and bbox_preds (B,Q,4), got {tuple(cls_scores.shape)} "
                         f"and {tuple(bbox_preds.shape)}")
    B, Q, C = cls_scores.shape
    assert len(img_metas) == B
    dev = cls_scores.device
    logits = cls_scores.detach().to(torch.float32).contiguous()
    boxes = bbox_preds.detach().to(torch.float32).contiguous()
    hw = _img_hw(img_metas, dev)
    lib = _lib.lib()
    ws = torch.empty(max(int(lib.semidetr_nms_workspace_bytes(B, Q, C)), 16), dtype=torch.uint8, device=dev)
    dets = torch.empty((B, max_per_img, 5), dtype=torch.float32, device=dev)
    labels = torch.empty((B, max_per_img), dtype=torch.int64, device=dev)
    count = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.semidetr_pseudo_nms_f32(_lib.current_stream_ptr(), _p(logits), _p(boxes), _p(hw), B, Q, C,
                                         float(score_thr), float(iou_threshold), int(max_per_img), _p(ws),
                                         ws.numel(), _p(dets), _p(labels), _p(count))
    _lib.check(rc, "semidetr_pseudo_nms_f32")
    return dets, labels, count


def get_bboxes_for_pseudo_label(cls_scores, bbox_preds, img_metas, score_thr=0.01, iou_threshold=0.6,
                                max_per_img=300):
    """cls_scores (B,Q,C) raw logits and bbox_preds (B,Q,4) normalised cxcywh of the LAST decoder layer
    (``all_cls_scores[-1]``, ``all_bbox_preds[-1]``); img_metas: dicts with 'img_shape'.
    Returns the reference's result_list: [(det_bboxes (k,5), det_labels (k,)), ...]."""
    dets, labels, count = _nms_batch(cls_scores, bbox_preds, img_metas, score_thr, iou_threshold, max_per_img)
    kept = count.tolist()              # the one host sync: list lengths are data dependent
    return [(dets[b, :kept[b]], labels[b, :kept[b]]) for b in range(len(kept))]


class PendingPseudoLabels:
    """Pseudo labels whose list lengths are still on their way to the host (pinned buffer + event): the kernels
    and the read-back are queued, ``result()`` waits for the event and slices.  Lets the step keep launching the
    student's forward while the teacher's boxes are being decoded."""

    def __init__(self, dets, labels, out_boxes, out_labels, out_scores, counts, max_per_img):
        self._t = (dets, labels, out_boxes, out_labels, out_scores)
        self._max = max_per_img
        self._host = torch.empty(tuple(counts.shape), dtype=torch.int32).pin_memory()
        self._host.copy_(counts, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()
        self._lists = None

    def result(self, return_proposals=False):
        if self._lists is None:
            self._event.synchronize()
            both = self._host.tolist()
            dets, labels, ob, ol, os_ = self._t
            m, B = self._max, dets.shape[0]
            self._lists = ([ob[b * m:b * m + both[1][b]] for b in range(B)],
                           [ol[b * m:b * m + both[1][b]] for b in range(B)],
                           [os_[b * m:b * m + both[1][b]] for b in range(B)],
                           [(dets[b, :both[0][b]], labels[b, :both[0][b]]) for b in range(B)])
        return self._lists if return_proposals else self._lists[:3]


def teacher_pseudo_labels(cls_scores, bbox_preds, img_metas, score_thr=0.01, iou_threshold=0.6, max_per_img=300,
                          return_proposals=False, wait=True):
    """extract_teacher_info's box path end to end on the device (dino_detr_ssod.py:904-939): decoding + NMS, then
    the mean+std filter, chained through device-side counts.  Returns (det_bboxes, det_labels, det_scores); with
    ``wait=False`` a ``PendingPseudoLabels`` whose ``result()`` gives the same lists later (one event wait)."""
    dets, labels, count = _nms_batch(cls_scores, bbox_preds, img_metas, score_thr, iou_threshold, max_per_img)
    B, dev = dets.shape[0], dets.device
    offs = _lib_small([b * max_per_img for b in range(B + 1)], torch.int32, dev)
    out_boxes = torch.empty((B * max_per_img, 4), dtype=torch.float32, device=dev)
    out_labels = torch.empty(B * max_per_img, dtype=torch.int64, device=dev)
    out_scores = torch.empty(B * max_per_img, dtype=torch.float32, device=dev)
    out_count = torch.empty(B, dtype=torch.int32, device=dev)
    out_thr = torch.empty(B, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib().semidetr_pseudo_label_filter_f32(
            _lib.current_stream_ptr(), _p(dets), _p(labels), _p(offs), _p(count), B, _p(out_boxes), _p(out_labels),
            _p(out_scores), _p(None), _p(out_count), _p(out_thr))
    _lib.check(rc, "semidetr_pseudo_label_filter_f32")
    pending = PendingPseudoLabels(dets, labels, out_boxes, out_labels, out_scores, torch.stack([count, out_count]),
                                  max_per_img)
    return pending.result(return_proposals) if wait else pending


def transform_bboxes(bbox, M, out_shape):
    """Transform2D.transform_bboxes for a list of (K_i, 4|5) boxes, (3,3) matrices and (h, w[, c]) shapes (or a
    single triple).  A 5th column (score) is passed through."""
    single = isinstance(bbox, torch.Tensor)
    boxes = [bbox] if single else list(bbox)
    mats = [M] if single else list(M)
    shapes = [out_shape] if single else list(out_shape)
    assert len(boxes) == len(mats) == len(shapes)
    B = len(boxes)
    if B == 0:
        return []
    dev = boxes[0].device
    if dev.type != "cuda":
        raise RuntimeError("transform_bboxes: tensors must live on the GPU (no CPU fallback)")
    counts = [int(b.shape[0]) for b in boxes]
    offs = [0]
    for c in counts:
        offs.append(offs[-1] + c)
    total = offs[-1]
    out_list = [b for b in boxes]
    if total:
        cat = torch.cat([b[:, :4].to(torch.float32) for b in boxes]).contiguous()
        mt = torch.stack([m.to(device=dev, dtype=torch.float32).reshape(3, 3) for m in mats]).contiguous()
        hw = _lib_small([[float(s[0]), float(s[1])] for s in shapes], torch.float32, dev)
        offs_dev = _lib_small(offs, torch.int32, dev)
        out = torch.empty((total, 4), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().semidetr_transform_bboxes_f32(_lib.current_stream_ptr(), _p(cat), 4, _p(offs_dev), _p(None),
                                                          B, max(counts), _p(mt), _p(hw), _p(out))
        _lib.check(rc, "semidetr_transform_bboxes_f32")
        out_list = []
        for b in range(B):
            o = out[offs[b]:offs[b + 1]].to(boxes[b].dtype)
            if counts[b] and boxes[b].shape[1] > 4:
                o = torch.cat([o, boxes[b][:, 4:]], dim=1)
            out_list.append(o if counts[b] else boxes[b])        # empty input is returned as is (bbox_utils.py:177-178)
    return out_list[0] if single else out_list
