"""Task-aligned focal loss of the warm-up stage on the MI355X (``csrc/tal_loss.hip``).

Mirrors ``TaskAlignedFocalLoss`` / ``task_aigned_focal_loss`` (detr_od/models/losses/task_aligned_focal_loss.py:35-66,
:136-200): same constructor, same ``forward(prob, target, alignment_metric, weight, avg_factor, reduction_override)``.
The reference evaluates ~10 elementwise torch kernels forward and as many backward per decoder layer; here the
loss sum and its gradient come out of one streaming pass.  ``forward_logits`` additionally folds the
``cls_scores.sigmoid()`` of the call site (dino_detr_ssod_head.py:693-694) into the same pass.
"""
import ctypes

import torch
from torch import nn

from . import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class _TalLossSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, target, metric, gamma, input_is_prob):
        if not x.is_cuda:
            raise RuntimeError("task_aligned_focal_loss: tensors must live on the GPU (no CPU fallback)")
        if x.dim() != 2 or target.shape != x.shape[:1] or metric.shape != x.shape[:1]:
            raise ValueError(f"expected (N,C) scores with (N,) target / alignment_metric, got {tuple(x.shape)}, "
                             f"{tuple(target.shape)}, {tuple(metric.shape)}")
        xin = x.detach().to(torch.float32).contiguous()
        tg = target.detach().to(device=x.device, dtype=torch.int64).contiguous()
        mt = metric.detach().to(device=x.device, dtype=torch.float32).contiguous()
        lib = _lib.lib()
        ws = torch.empty(int(lib.semidetr_tal_loss_workspace_bytes()), dtype=torch.uint8, device=x.device)
        out = torch.empty(1, dtype=torch.float32, device=x.device)
        grad = torch.empty_like(xin) if x.requires_grad else None
        with torch.cuda.device(x.device):
            rc = lib.semidetr_tal_loss_f32(_lib.current_stream_ptr(), _p(xin), _p(tg), _p(mt), xin.shape[0], xin.shape[1],
                                           float(gamma), int(bool(input_is_prob)), _p(ws), _p(out), _p(grad))
        _lib.check(rc, "semidetr_tal_loss_f32")
        ctx.save_for_backward(grad)
        ctx.in_dtype = x.dtype
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g).to(ctx.in_dtype), None, None, None, None


def task_aligned_focal_loss(scores, target, alignment_metric, weight=None, gamma=2.0, reduction="mean", avg_factor=None,
                            from_logits=False):
    """``task_aigned_focal_loss`` (task_aligned_focal_loss.py:35-66).  ``scores`` are probabilities, or raw logits
    with ``from_logits=True`` (the sigmoid is then fused).  ``weight`` must be None (the reference's call site passes
    none) and ``reduction`` 'mean' or 'sum' -- the element-wise loss tensor is never materialised."""
    if weight is not None:
        raise NotImplementedError("task_aligned_focal_loss: element-wise weights are not built (no caller in the reference)")
    total = _TalLossSum.apply(scores, target, alignment_metric, gamma, not from_logits)
    if avg_factor is None:                       # mmdet weight_reduce_loss (losses/utils.py:29-55)
        if reduction == "mean":
            return total / max(scores.numel(), 1)
        if reduction == "sum":
            return total
        raise NotImplementedError("task_aligned_focal_loss: reduction='none' would materialise the element-wise loss")
    if reduction == "mean":
        return total / avg_factor
    if reduction == "none":
        raise NotImplementedError("task_aligned_focal_loss: reduction='none' would materialise the element-wise loss")
    raise ValueError('avg_factor can not be used with reduction="sum"')


class TaskAlignedFocalLoss(nn.Module):
    def __init__(self, use_sigmoid=True, gamma=2.0, reduction="mean", loss_weight=1.0):
        super().__init__()
        self.use_sigmoid = use_sigmoid
        self.gamma = gamma
        self.reduction = reduction
        self.loss_weight = loss_weight

    def _call(self, scores, target, alignment_metric, weight, avg_factor, reduction_override, from_logits):
        assert reduction_override in (None, "none", "mean", "sum")
        reduction = reduction_override if reduction_override else self.reduction
        if not self.use_sigmoid:
            raise NotImplementedError
        return self.loss_weight * task_aligned_focal_loss(scores, target, alignment_metric, weight, gamma=self.gamma,
                                                          reduction=reduction, avg_factor=avg_factor,
                                                          from_logits=from_logits)

    def forward(self, prob, target, alignment_metric, weight=None, avg_factor=None, reduction_override=None):
        return self._call(prob, target, alignment_metric, weight, avg_factor, reduction_override, False)

    def forward_logits(self, cls_scores, target, alignment_metric, weight=None, avg_factor=None, reduction_override=None):
        """Same loss on the raw logits: replaces ``self.loss_cls1(cls_scores.sigmoid(), ...)`` (head.py:693-694)."""
        return self._call(cls_scores, target, alignment_metric, weight, avg_factor, reduction_override, True)
