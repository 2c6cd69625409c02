"""``MSDeformAttnFunction`` -- autograd boundary of the operator.

Same call contract as the reference (detr_od/models/utils/ops/functions/ms_deform_attn_func.py:21-38):
``apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
im2col_step)``; backward is once-differentiable and returns
``(grad_value, None, None, grad_sampling_loc, grad_attn_weight, None)``.
The reference's pure-PyTorch debug routine (``ms_deform_attn_core_pytorch``, :41-61) is deliberately
NOT part of this package: the product has no second implementation to fall back to; the CPU checker
lives under ``oracle/`` and is test infrastructure.
"""
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ... import MultiScaleDeformableAttention as MSDA


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        output = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                             sampling_locations, attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights = ctx.saved_tensors
        grad_value, grad_sampling_loc, grad_attn_weight = MSDA.ms_deform_attn_backward(
            value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
            grad_output.contiguous(), ctx.im2col_step)
        return grad_value, None, None, grad_sampling_loc, grad_attn_weight, None
