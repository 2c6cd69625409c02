"""CPU oracle for the Semi-DETR hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package; the product (``semi-detr_amd``) never does.  It wraps ``oracle/_build/liboracle.so`` (plain C,
see msda_oracle.c / lsap_oracle.c / hotpath_oracle.c, each citing the reference file:line it
restates) behind numpy-in / numpy-out functions.

Parity pin: every function here is checked against fixtures under ``tests/golden`` that were produced
by ``oracle/gen_golden.py`` from the reference's own Python (``ms_deform_attn_core_pytorch``, mmdet
match costs, ``bbox_overlaps``) imported by path in the build container, and against
``scipy.optimize.linear_sum_assignment`` (scipy 1.15.3) for the assignment solver.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")


def build(force=False):
    """Compile the C restatement with gcc (a few seconds).  Building the checker is not using it.  `make` decides
    what is stale from oracle/Makefile's own source list, so no second list can drift from it."""
    subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.ema_momentum_oracle.restype = ctypes.c_double
        _lib.ema_momentum_oracle.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_int64]
    return _lib


def _p(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _level_start(shapes):
    hw = shapes[:, 0] * shapes[:, 1]
    return np.concatenate([[0], np.cumsum(hw)[:-1]]).astype(np.int64)


def msda_forward(value, shapes, loc, attn, level_start=None):
    """value (N,S,M,D), shapes (L,2) int64 [(H,W)], loc (N,Lq,M,L,P,2), attn (N,Lq,M,L,P) -> (N,Lq,M*D)."""
    dt = np.float64 if value.dtype == np.float64 else np.float32
    value, loc, attn = _c(value, dt), _c(loc, dt), _c(attn, dt)
    shapes = _c(shapes, np.int64)
    starts = _level_start(shapes) if level_start is None else _c(level_start, np.int64)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.empty((N, Lq, M * D), dt)
    fn = lib().msda_oracle_forward_f64 if dt == np.float64 else lib().msda_oracle_forward_f32
    fn(_p(value), _p(shapes), _p(starts), _p(loc), _p(attn), N, S, M, D, L, Lq, P, _p(out))
    return out


def msda_backward(value, shapes, loc, attn, grad_out, level_start=None, parallel=False):
    """-> grad_value, grad_loc, grad_attn (same shapes/dtype as value, loc, attn).
    parallel=True runs the (image, head, level) slices of grad_value as independent OpenMP tasks (no atomics; the
    result is bit-identical to the serial default, tests/test_oracle_msda.py checks that)."""
    dt = np.float64 if value.dtype == np.float64 else np.float32
    value, loc, attn, grad_out = _c(value, dt), _c(loc, dt), _c(attn, dt), _c(grad_out, dt)
    shapes = _c(shapes, np.int64)
    starts = _level_start(shapes) if level_start is None else _c(level_start, np.int64)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    gv, gl, ga = np.empty_like(value), np.empty_like(loc), np.empty_like(attn)
    name = "msda_oracle_backward_" + ("omp_" if parallel else "") + ("f64" if dt == np.float64 else "f32")
    fn = getattr(lib(), name)
    fn(_p(value), _p(shapes), _p(starts), _p(loc), _p(attn), _p(grad_out), N, S, M, D, L, Lq, P,
       _p(gv), _p(gl), _p(ga))
    return gv, gl, ga


class LsapError(ValueError):
    pass


def lsap(cost):
    """Rectangular LSAP with scipy's exact tie behaviour.  cost (nr,nc) any float dtype (up-cast to f64)."""
    c = _c(cost, np.float64)
    nr, nc = c.shape
    k = min(nr, nc)
    a, b = np.zeros(k, np.int64), np.zeros(k, np.int64)
    rc = lib().lsap_oracle_solve(ctypes.c_int64(nr), ctypes.c_int64(nc), _p(c), _p(a), _p(b))
    if rc == -1:
        raise LsapError("cost matrix is infeasible")
    if rc == -2:
        raise LsapError("matrix contains invalid numeric entries")
    return a, b


def match_cost(bbox_pred, cls_pred, gt_bboxes, gt_labels, img_w, img_h, w_cls=2.0, alpha=0.25,
               gamma=2.0, eps=1e-12, w_reg=5.0, box_format="xywh", w_iou=2.0, iou_mode="giou",
               parts=False):
    bbox_pred, cls_pred = _c(bbox_pred, np.float32), _c(cls_pred, np.float32)
    gt_bboxes, gt_labels = _c(gt_bboxes, np.float32).reshape(-1, 4), _c(gt_labels, np.int64)
    Q, C, G = bbox_pred.shape[0], cls_pred.shape[1], gt_bboxes.shape[0]
    outs = [np.zeros((Q, G), np.float32) for _ in range(4)]
    f = ctypes.c_float
    lib().match_cost_oracle(_p(bbox_pred), _p(cls_pred), _p(gt_bboxes), _p(gt_labels), Q, C, G,
                            f(img_w), f(img_h), f(w_cls), f(alpha), f(gamma), f(eps), f(w_reg),
                            int(box_format == "xywh"), f(w_iou), int(iou_mode == "giou"),
                            *[_p(o) for o in outs])
    return tuple(outs) if parts else outs[0]


def assign_scatter(rows, cols, gt_labels, Q):
    rows, cols, gt_labels = _c(rows, np.int64), _c(cols, np.int64), _c(gt_labels, np.int64)
    gi, lab = np.empty(Q, np.int64), np.empty(Q, np.int64)
    lib().assign_scatter_oracle(_p(rows), _p(cols), len(rows), _p(gt_labels), Q, _p(gi), _p(lab))
    return gi, lab


def hungarian_assign(bbox_pred, cls_pred, gt_bboxes, gt_labels, img_w, img_h, **kw):
    """hungarian_assigner.py:55-188 end to end -> (assigned_gt_inds, assigned_labels, rows, cols)."""
    Q, G = len(bbox_pred), len(gt_bboxes)
    gi, lab = np.full(Q, -1, np.int64), np.full(Q, -1, np.int64)
    e = np.zeros(0, np.int64)
    if G == 0 or Q == 0:
        if G == 0:
            gi[:] = 0
        return gi, lab, e, e
    rows, cols = lsap(match_cost(bbox_pred, cls_pred, gt_bboxes, gt_labels, img_w, img_h, **kw))
    gi, lab = assign_scatter(rows, cols, gt_labels, Q)
    return gi, lab, rows, cols


def ema_momentum(momentum, warm_up, step):
    return lib().ema_momentum_oracle(float(momentum), float(warm_up), int(step))


def ema_update(teacher, student, momentum):
    """In place on a float32 numpy array ``teacher``; returns it."""
    assert teacher.dtype == np.float32 and teacher.flags.c_contiguous
    student = _c(student, np.float32)
    lib().ema_oracle(_p(teacher), _p(student), ctypes.c_int64(teacher.size), ctypes.c_double(momentum))
    return teacher


def pseudo_label_filter(proposal):
    """proposal (K,5) -> (keep_idx int64 ascending, thr float32)."""
    proposal = _c(proposal, np.float32).reshape(-1, 5)
    K = proposal.shape[0]
    keep = np.zeros(max(K, 1), np.int64)
    thr = ctypes.c_float(0)
    n = lib().pseudo_label_oracle(_p(proposal), K, _p(keep), ctypes.byref(thr))
    return keep[:n].copy(), np.float32(thr.value)


def pseudo_nms(cls_logits, bbox_pred, img_h, img_w, score_thr=0.01, iou_thr=0.6, max_num=300):
    """One image: cls_logits (Q,C), bbox_pred (Q,4) -> (dets (k,5) float32, labels (k,) int64).
    nms_oracle.c:pseudo_nms_oracle (head.py:1364-1395 + bbox_nms.py:8-95 + mmcv batched_nms restated)."""
    cls_logits = _c(cls_logits, np.float32)
    bbox_pred = _c(bbox_pred, np.float32)
    Q, C = cls_logits.shape
    cap = max(Q * C, 1) if max_num <= 0 else max_num
    dets = np.zeros((cap, 5), np.float32)
    labels = np.zeros(cap, np.int64)
    fn = lib().pseudo_nms_oracle
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                   ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    n = fn(_p(cls_logits), _p(bbox_pred), Q, C, float(img_h), float(img_w), float(score_thr), float(iou_thr),
           int(max_num), _p(dets), _p(labels))
    return dets[:n].copy(), labels[:n].copy()


def transform_bboxes(boxes, M, out_h, out_w):
    """boxes (K,4), M (3,3) -> (K,4); nms_oracle.c:transform_bboxes_oracle (bbox_utils.py:167-192)."""
    boxes = _c(boxes, np.float32).reshape(-1, 4)
    M = _c(M, np.float32).reshape(9)
    out = np.zeros_like(boxes)
    fn = lib().transform_bboxes_oracle
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    fn.restype = None
    fn(_p(boxes), boxes.shape[0], _p(M), float(out_h), float(out_w), _p(out))
    return out


def o2m_assign(bbox_pred, cls_prob, gt_bboxes, gt_labels, img_w, img_h, topk=13, alpha=1.0, beta=6.0, dynamic_k=False):
    """One image -> (gt_inds int64, labels int64, max_overlaps f32, assign_metrics f32); o2m_oracle.c
    (o2m_assigner.py:50-170)."""
    bbox_pred, cls_prob = _c(bbox_pred, np.float32), _c(cls_prob, np.float32)
    gt_bboxes, gt_labels = _c(gt_bboxes, np.float32).reshape(-1, 4), _c(gt_labels, np.int64)
    Q, C, G = bbox_pred.shape[0], cls_prob.shape[1], gt_bboxes.shape[0]
    gi, lab = np.zeros(max(Q, 1), np.int64), np.zeros(max(Q, 1), np.int64)
    mo, am = np.zeros(max(Q, 1), np.float32), np.zeros(max(Q, 1), np.float32)
    fn = lib().o2m_assign_oracle
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                                ctypes.c_float, ctypes.c_float] + [ctypes.c_void_p] * 4
    fn(_p(bbox_pred), _p(cls_prob), _p(gt_bboxes), _p(gt_labels), Q, C, G, float(img_w), float(img_h), int(topk),
       int(bool(dynamic_k)), float(alpha), float(beta), _p(gi), _p(lab), _p(mo), _p(am))
    return gi[:Q], lab[:Q], mo[:Q], am[:Q]


def o2m_targets(gt_inds, max_overlaps, assign_metrics, gt_bboxes, gt_labels, img_w, img_h, num_classes):
    """-> (labels_full int64, bbox_targets (Q,4) f32, norm_metrics f32); o2m_oracle.c (head.py:1114-1165)."""
    gt_inds, max_overlaps = _c(gt_inds, np.int64), _c(max_overlaps, np.float32)
    assign_metrics = _c(assign_metrics, np.float32)
    gt_bboxes, gt_labels = _c(gt_bboxes, np.float32).reshape(-1, 4), _c(gt_labels, np.int64)
    Q, G = gt_inds.shape[0], gt_bboxes.shape[0]
    lf, bt, nm = np.zeros(max(Q, 1), np.int64), np.zeros((max(Q, 1), 4), np.float32), np.zeros(max(Q, 1), np.float32)
    fn = lib().o2m_targets_oracle
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int64] + \
                  [ctypes.c_void_p] * 3
    fn(_p(gt_inds), _p(max_overlaps), _p(assign_metrics), _p(gt_bboxes), _p(gt_labels), Q, G, float(img_w), float(img_h),
       int(num_classes), _p(lf), _p(bt), _p(nm))
    return lf[:Q], bt[:Q], nm[:Q]


def tal_loss(scores, labels, metrics, gamma=2.0, input_is_prob=True, want_grad=True):
    """-> (loss_sum float64, grad (N,C) f32 or None); o2m_oracle.c:tal_loss_oracle (task_aligned_focal_loss.py:35-66)."""
    scores = _c(scores, np.float32)
    labels, metrics = _c(labels, np.int64), _c(metrics, np.float32)
    N, C = scores.shape
    grad = np.zeros_like(scores) if want_grad else None
    fn = lib().tal_loss_oracle
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_void_p]
    s = fn(_p(scores), _p(labels), _p(metrics), N, C, float(gamma), int(bool(input_is_prob)), _p(grad))
    return s, grad
