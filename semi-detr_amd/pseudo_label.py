"""Teacher pseudo-label filter on the MI355X (``csrc/pseudo_label.hip``).

Replaces the per-image Python loop of ``DinoDetrSSOD.extract_teacher_info``
(detr_ssod/models/dino_detr_ssod.py:918-939): threshold = mean + unbiased std of the image's scores, keep
``score >= thr``, drop boxes with non-positive width/height, keep post-NMS order.  One launch and one
host read-back of the kept counts for the whole batch (the reference syncs several times per image).
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
            _lib.current_stream_ptr(), _p(prop), _p(lab), _p(offs_dev), B, _p(out_boxes), _p(out_labels),
            _p(out_scores), _p(out_keep), _p(out_count), _p(out_thr))
    _lib.check(rc, "semidetr_pseudo_label_filter_f32")
    kept = out_count.tolist()          # the one host sync: list lengths are data dependent
    det_bboxes = [out_boxes[offs[b]:offs[b] + kept[b]] for b in range(B)]
    det_labels = [out_labels[offs[b]:offs[b] + kept[b]].to(proposal_label_list[b].dtype) for b in range(B)]
    det_scores = [out_scores[offs[b]:offs[b] + kept[b]] for b in range(B)]
    if return_threshold:
        return det_bboxes, det_labels, det_scores, out_thr
    return det_bboxes, det_labels, det_scores
