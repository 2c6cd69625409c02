"""``MSDeformAttn`` -- the nn.Module around the operator.

Drop-in for detr_od/models/utils/ops/modules/ms_deform_attn.py:30-126: same constructor signature,
same sub-module names (``sampling_offsets``, ``attention_weights``, ``value_proj``, ``output_proj`` ->
identical ``state_dict`` keys, so published DINO / Semi-DETR checkpoints load unchanged), same
initialisation (:62-76), same ``forward`` arguments, errors and arithmetic (:78-126).  The four Linear
layers are dense GEMMs and stay on hipBLASLt/MFMA through torch; the sampling + aggregation and its
gradient go through ``MSDeformAttnFunction`` to the hand-written gfx950 kernels.
"""
import itertools
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn

from ... import MultiScaleDeformableAttention as MSDA
from ..functions import MSDeformAttnFunction, MSDeformAttnFusedFunction


def _is_power_of_2(n):
    if not isinstance(n, int) or n < 0:
        raise ValueError("invalid input for _is_power_of_2: {} (type: {})".format(n, type(n)))
    return n != 0 and (n & (n - 1)) == 0


_policy_slots = itertools.count(1)


def _next_policy_slot():
    return (next(_policy_slots) - 1) % 255 + 1


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if not _is_power_of_2(d_model // n_heads):
            warnings.warn("MSDeformAttn: a per-head dimension that is not a power of 2 (here {}) takes the "
                          "generic kernel; 32 channels per head is the tuned gfx950 path."
                          .format(d_model // n_heads))
        self.im2col_step = 64        # kept for config/state compatibility; one launch covers the batch
        # Fuse softmax + location arithmetic (and their backward) into the sampling kernels whenever the
        # configuration allows it (fp32, 32 channels per head): same results, none of the (N,Lq,M,L,P[,2])
        # intermediates in HBM.  Not a parameter / buffer -> state_dict is unchanged.  Set False for the
        # reference's op-by-op sequence.
        self.fuse_prologue = True
        # Which of the two encoder forward kernels runs follows how far THIS instance's learned offsets reach (semidetr_hip.h:
        # SEMIDETR_MSDA_POLICY_SLOT): instances take consecutive slots 1..255 (the reference builds 12 per model,
        # transformer.py:609,760; beyond 255 instances slots are shared, which only mixes their counts).  Plain attribute.
        self.policy_slot = _next_policy_slot()
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points

        self.sampling_offsets = nn.Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = nn.Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = nn.Linear(d_model, d_model)
        self.output_proj = nn.Linear(d_model, d_model)
        self._reset_parameters()

    def __setstate__(self, state):
        # copy.deepcopy (an EMA teacher cloned from the student) and unpickling (torch.save(model), checkpoints that pickle modules)
        # both come through here: the copy is its own call site with its own learned offsets, so it takes a FRESH slot -- sharing
        # the original's would mix the two instances' sample-spread counts; a module pickled before the attribute existed gets one too
        super().__setstate__(state)
        self.__dict__["policy_slot"] = _next_policy_slot()
        self.__dict__.setdefault("fuse_prologue", True)

    def _reset_parameters(self):
        # ms_deform_attn.py:62-76 -- head m looks along direction 2*pi*m/M, point i at distance i+1
        nn.init.constant_(self.sampling_offsets.weight.data, 0.0)
        thetas = torch.arange(self.n_heads, dtype=torch.float32) * (2.0 * math.pi / self.n_heads)
        grid = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid = grid / grid.abs().max(-1, keepdim=True)[0]
        grid = grid.view(self.n_heads, 1, 1, 2).repeat(1, self.n_levels, self.n_points, 1)
        grid = grid * torch.arange(1, self.n_points + 1, dtype=torch.float32).view(1, 1, -1, 1)
        with torch.no_grad():
            self.sampling_offsets.bias = nn.Parameter(grid.reshape(-1))
        nn.init.constant_(self.attention_weights.weight.data, 0.0)
        nn.init.constant_(self.attention_weights.bias.data, 0.0)
        nn.init.xavier_uniform_(self.value_proj.weight.data)
        nn.init.constant_(self.value_proj.bias.data, 0.0)
        nn.init.xavier_uniform_(self.output_proj.weight.data)
        nn.init.constant_(self.output_proj.bias.data, 0.0)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index,
                input_padding_mask=None):
        """query (N, Lq, C); reference_points (N, Lq, L, 2|4) in [0,1]; input_flatten (N, sum HW, C);
        input_spatial_shapes (L, 2) [(H, W)]; input_level_start_index (L,); input_padding_mask (N, sum HW)
        True = padding.  Returns (N, Lq, C)."""
        N, Len_q, _ = query.shape
        N, Len_in, _ = input_flatten.shape
        # The reference asserts `(shapes[:, 0] * shapes[:, 1]).sum() == Len_in` on a device tensor (ms_deform_attn.py:90):
        # one blocking device-to-host copy per layer call, ~60 per SSOD step.  Same check, but answered from a cache keyed
        # on the (spatial_shapes, level_start_index) tensors -- all layers of a forward pass share them, so at most one
        # synchronisation per step, none when the caller keeps the two tensors alive between steps.
        assert MSDA.pyramid_check(input_spatial_shapes, input_level_start_index, Len_in) & 1
        M, L, P = self.n_heads, self.n_levels, self.n_points

        value = self.value_proj(input_flatten).view(N, Len_in, M, self.d_model // M)
        offsets = self.sampling_offsets(query).view(N, Len_q, M, L, P, 2)
        logits = self.attention_weights(query).view(N, Len_q, M, L * P)
        if reference_points.shape[-1] not in (2, 4):
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead."
                             .format(reference_points.shape[-1]))
        if self.fuse_prologue and MSDA.fused_supported(value, reference_points, offsets, logits):
            # the padding mask (ms_deform_attn.py:95-96) goes INTO the kernels: no masked copy of `value`, no masking of its
            # gradient
            output = MSDeformAttnFusedFunction.apply(value.contiguous(), input_spatial_shapes,
                                                     input_level_start_index, reference_points.contiguous(),
                                                     offsets.contiguous(), logits.contiguous(), input_padding_mask,
                                                     getattr(self, "policy_slot", 0))
            return self.output_proj(output)
        if input_padding_mask is not None:
            value = value.masked_fill(input_padding_mask[..., None, None], float(0))
        weights = F.softmax(logits, -1).view(N, Len_q, M, L, P)
        if reference_points.shape[-1] == 2:
            normalizer = torch.stack([input_spatial_shapes[..., 1], input_spatial_shapes[..., 0]], -1)
            locations = reference_points[:, :, None, :, None, :] + offsets / normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            locations = reference_points[:, :, None, :, None, :2] \
                + offsets / P * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead."
                             .format(reference_points.shape[-1]))

        if value.dtype == torch.float16:      # amp: the op itself runs in fp32 (ms_deform_attn.py:114-120)
            output = MSDeformAttnFunction.apply(value.float(), input_spatial_shapes, input_level_start_index,
                                                locations.float(), weights.float(), self.im2col_step, getattr(self, "policy_slot", 0))
            return self.output_proj(output.to(torch.float16))
        output = MSDeformAttnFunction.apply(value.contiguous(), input_spatial_shapes, input_level_start_index,
                                            locations.contiguous(), weights.contiguous(), self.im2col_step, getattr(self, "policy_slot", 0))
        return self.output_proj(output)
