"""``MSDeformAttnFunction`` -- autograd boundary of the operator.

Same call contract as the reference (detr_od/models/utils/ops/functions/ms_deform_attn_func.py:21-38):
``apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights,
im2col_step)``; backward is once-differentiable and returns
``(grad_value, None, None, grad_sampling_loc, grad_attn_weight, None)``.
The reference's pure-PyTorch debug routine (``ms_deform_attn_core_pytorch``, :41-61) is deliberately
NOT part of this package: the product has no second implementation to fall back to; the CPU checker
lives under ``oracle/`` and is test infrastructure.
"""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ... import MultiScaleDeformableAttention as MSDA


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations,
                attention_weights, im2col_step, policy_slot=0):
        # policy_slot (optional, not in the reference's signature): the call site's slot of the encoder forward-kernel choice
        ctx.im2col_step = im2col_step
        output = MSDA.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index,
                                             sampling_locations, attention_weights, ctx.im2col_step, policy_slot)
        # The backward's small-gradient kernel follows what the call site's counts said NOW, not whenever the backward runs (the two
        # gathers agree to fp32 rounding only, and the reference's grad_sampling_loc / grad_attn_weight are deterministic): ADVICE r05
        ctx.policy_slot = policy_slot | MSDA.gather_choice(policy_slot)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                              attention_weights)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, spatial_shapes, level_start_index, sampling_locations, attention_weights = ctx.saved_tensors
        grad_value, grad_sampling_loc, grad_attn_weight = MSDA.ms_deform_attn_backward(
            value, spatial_shapes, level_start_index, sampling_locations, attention_weights,
            grad_output.contiguous(), ctx.im2col_step, ctx.policy_slot)
        return (grad_value, None, None, grad_sampling_loc, grad_attn_weight, None, None)[:len(ctx.needs_input_grad)]


class MSDeformAttnFusedFunction(Function):
    """MSDeformAttn.forward between the Linear layers as ONE op: consumes the
    reference points, the raw sampling offsets and the raw attention logits; softmax, location arithmetic
    (detr_od/models/utils/ops/modules/ms_deform_attn.py:99-111) and their backward run inside the gfx950 kernels.
    ``apply(value, spatial_shapes, level_start_index, reference_points, sampling_offsets, attn_logits[, padding_mask[, policy_slot]])``.
    ``padding_mask`` (N, S) bool, True = padding: ``value.masked_fill(mask[..., None], 0)`` (ms_deform_attn.py:95-96) folded
    into the kernels -- pass the UNMASKED value; its gradient comes back with zero rows at the padded pixels."""

    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, reference_points, sampling_offsets, attn_logits,
                padding_mask=None, policy_slot=0):
        if padding_mask is not None:
            padding_mask = padding_mask.contiguous()
        output = MSDA.ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index, reference_points,
                                                   sampling_offsets, attn_logits, padding_mask, policy_slot)
        ctx.policy_slot = policy_slot | MSDA.gather_choice(policy_slot)      # (see MSDeformAttnFunction.forward)
        ctx.save_for_backward(value, spatial_shapes, level_start_index, reference_points, sampling_offsets,
                              attn_logits, padding_mask)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, ref, off, logits, padding_mask = ctx.saved_tensors
        grad_value, grad_off, grad_logits = MSDA.ms_deform_attn_fused_backward(
            value, shapes, starts, ref, off, logits, grad_output.contiguous(), padding_mask, ctx.policy_slot)
        grad_ref = None
        if ctx.needs_input_grad[3]:
            # d loc / d ref: the chain rule through the (elementwise) location arithmetic, from grad_off
            L, P = off.shape[3], off.shape[4]
            if ref.shape[-1] == 2:      # loc = ref + off / (W, H)  ->  d/d ref = sum_{m,p} grad_off * (W, H)
                norm = torch.stack([shapes[:, 1], shapes[:, 0]], -1).to(off.dtype)          # (L, 2)
                grad_ref = (grad_off * norm[None, None, None, :, None, :]).sum((2, 4))
            else:                       # loc = ref_xy + off / P * ref_wh * 0.5
                wh = ref[:, :, None, :, None, 2:]
                scale = 0.5 * wh / P
                grad_loc = torch.where(scale != 0, grad_off / scale, torch.zeros_like(grad_off))
                grad_ref = torch.cat([grad_loc.sum((2, 4)), (grad_loc * off * (0.5 / P)).sum((2, 4))], -1)
        return (grad_value, None, None, grad_ref, grad_off, grad_logits, None, None)[:len(ctx.needs_input_grad)]
