"""semi-detr_amd: MI355X (gfx950) native hot path of Semi-DETR / DINO-DETR.

Scope: multi-scale deformable attention forward/backward, the Hungarian matcher
(cost matrix + LSAP + scatter), the mean-teacher EMA update and the pseudo-label filter, plus the
image-sharded data-parallel wrapper.  Host side is Python mirroring the reference's operator/plugin
surface; all arithmetic runs in hand-written HIP kernels behind the C ABI of
``include/semidetr_hip.h`` (``csrc/libsemidetr_hip.so``).  There is no CPU fallback.
"""
import sys as _sys

from . import _lib  # noqa: F401
from . import MultiScaleDeformableAttention as _msda_native

# The reference does `import MultiScaleDeformableAttention as MSDA`
# (detr_od/models/utils/ops/functions/ms_deform_attn_func.py:18); make that import resolve to ours.
_sys.modules.setdefault("MultiScaleDeformableAttention", _msda_native)

from .ops.functions import MSDeformAttnFunction, MSDeformAttnFusedFunction  # noqa: E402,F401
from .ops.modules import MSDeformAttn  # noqa: E402,F401
from .matcher import (AssignResult, BBoxL1Cost, FocalLossCost, HungarianAssigner, IoUCost,  # noqa: E402,F401
                      O2MAssigner, O2MAssignResult, linear_sum_assignment)
from .losses import TaskAlignedFocalLoss, task_aligned_focal_loss  # noqa: E402,F401
from .mean_teacher import MeanTeacher, ema_momentum, ema_update_, ema_update_flat_  # noqa: E402,F401
from .targets import TargetAssigner, get_targets, get_targets_layers  # noqa: E402,F401
from .pseudo_label import (filter_pseudo_labels, get_bboxes_for_pseudo_label, teacher_pseudo_labels,  # noqa: E402,F401
                           transform_bboxes)

__all__ = ["MSDeformAttnFunction", "MSDeformAttnFusedFunction", "MSDeformAttn", "HungarianAssigner", "FocalLossCost", "BBoxL1Cost",
           "IoUCost", "AssignResult", "O2MAssigner", "O2MAssignResult", "TaskAlignedFocalLoss", "task_aligned_focal_loss", "linear_sum_assignment", "MeanTeacher", "ema_momentum", "ema_update_",
           "ema_update_flat_", "filter_pseudo_labels", "get_bboxes_for_pseudo_label", "teacher_pseudo_labels",
           "transform_bboxes", "TargetAssigner", "get_targets", "get_targets_layers"]
