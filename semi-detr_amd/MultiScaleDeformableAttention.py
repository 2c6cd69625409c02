"""Python-level stand-in for the reference's pybind module ``MultiScaleDeformableAttention``.

Same two entry points, same argument order, same return values and the same precondition errors as
  src/vision.cpp:13-16  ->  src/ms_deform_attn.h:20-61  ->  src/cuda/ms_deform_attn_cuda.cu:20-153
but the work is done by the hand-written gfx950 kernels in ``csrc/msda.hip`` through the C ABI
(``semidetr_msda_{forward,backward}_{f32,f64}``).  Tensors are borrowed; outputs are freshly allocated
on the inputs' device; launches go to the current stream of that device; nothing synchronises.
"""
import ctypes

import torch

from . import _lib


_SCALAR_NAMES = {torch.float16: "Half", torch.bfloat16: "BFloat16", torch.int32: "Int", torch.int64: "Long",
                 torch.uint8: "Byte", torch.int8: "Char", torch.int16: "Short", torch.bool: "Bool"}


def _assert(cond, msg):
    if not cond:
        raise RuntimeError(msg)   # what AT_ASSERTM raises on the Python side


def _check_inputs(named, grad_output=None):
    value = named[0][1]
    # ms_deform_attn.h:27-38 -- the CPU path of the reference only raises
    _assert(value.is_cuda, "Not implemented on the CPU")
    tensors = list(named) + ([("grad_output", grad_output)] if grad_output is not None else [])
    for name, t in tensors:     # ms_deform_attn_cuda.cu:28-38, :93-105
        _assert(t.is_contiguous(), f"{name} tensor has to be contiguous")
    for name, t in tensors:
        _assert(t.is_cuda, f"{name} must be a CUDA tensor")
        _assert(t.device == value.device, f"{name} must be on the same device as value")


def _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step, what):
    _assert(value.dim() == 4 and sampling_loc.dim() == 6 and attn_weight.dim() == 5,
            f"{what}: expected value (N,S,M,D), sampling_loc (N,Lq,M,L,P,2), attn_weight (N,Lq,M,L,P)")
    batch, spatial_size, num_heads, channels = value.shape
    num_levels = spatial_shapes.shape[0]
    num_query, num_point = sampling_loc.shape[1], sampling_loc.shape[4]
    step = min(batch, int(im2col_step))                     # ms_deform_attn_cuda.cu:50-52
    _assert(step > 0 and batch % step == 0, f"batch({batch}) must divide im2col_step({step})")
    # the reference reads .data<int64_t>() of both index tensors -> anything but int64 raises
    _assert(spatial_shapes.dtype == torch.int64, "expected scalar type Long for spatial_shapes")
    _assert(level_start_index.dtype == torch.int64, "expected scalar type Long for level_start_index")
    _assert(value.dtype in (torch.float32, torch.float64),
            f'"{what}" not implemented for \'{_SCALAR_NAMES.get(value.dtype, str(value.dtype))}\'')
    _assert(sampling_loc.dtype == value.dtype and attn_weight.dtype == value.dtype,
            f"{what}: value, sampling_loc and attn_weight must share one dtype")
    _assert(tuple(sampling_loc.shape) == (batch, num_query, num_heads, num_levels, num_point, 2)
            and tuple(attn_weight.shape) == (batch, num_query, num_heads, num_levels, num_point)
            and level_start_index.numel() == num_levels and tuple(spatial_shapes.shape) == (num_levels, 2),
            f"{what}: inconsistent tensor shapes")
    return batch, spatial_size, num_heads, channels, num_levels, num_query, num_point


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    """-> Tensor (N, Lq, M*D).  Mirrors ms_deform_attn_cuda_forward (ms_deform_attn_cuda.cu:20-80)."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight)])
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                 im2col_step, "ms_deform_attn_forward_cuda")
    lib = _lib.lib()
    fn = lib.semidetr_msda_forward_f64 if value.dtype == torch.float64 else lib.semidetr_msda_forward_f32
    with torch.cuda.device(value.device):
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)  # kernel writes all of it
        if out.numel() == 0 or value.numel() == 0:
            return out.zero_()
        rc = fn(_lib.current_stream_ptr(), _p(value), _p(spatial_shapes), _p(level_start_index),
                _p(sampling_loc), _p(attn_weight), N, S, M, D, L, Lq, P, _p(out))
    _lib.check(rc, "ms_deform_attn_forward")
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step):
    """-> [grad_value, grad_sampling_loc, grad_attn_weight].  Mirrors ms_deform_attn_cuda_backward
    (ms_deform_attn_cuda.cu:83-153)."""
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes),
                   ("level_start_index", level_start_index), ("sampling_loc", sampling_loc),
                   ("attn_weight", attn_weight)], grad_output)
    N, S, M, D, L, Lq, P = _dims(value, spatial_shapes, level_start_index, sampling_loc, attn_weight,
                                 im2col_step, "ms_deform_attn_backward_cuda")
    _assert(grad_output.dtype == value.dtype and grad_output.numel() == N * Lq * M * D,
            "ms_deform_attn_backward_cuda: grad_output must be (N, Lq, M*D) of value's dtype")
    lib = _lib.lib()
    fn = lib.semidetr_msda_backward_f64 if value.dtype == torch.float64 else lib.semidetr_msda_backward_f32
    with torch.cuda.device(value.device):
        grad_value = torch.empty_like(value)          # zero-filled inside the call, on the same stream
        grad_loc = torch.empty_like(sampling_loc)     # fully written by the kernel
        grad_attn = torch.empty_like(attn_weight)
        if value.numel() == 0 or grad_loc.numel() == 0:
            return [grad_value.zero_(), grad_loc.zero_(), grad_attn.zero_()]
        rc = fn(_lib.current_stream_ptr(), _p(grad_output), _p(value), _p(spatial_shapes),
                _p(level_start_index), _p(sampling_loc), _p(attn_weight), N, S, M, D, L, Lq, P,
                _p(grad_value), _p(grad_loc), _p(grad_attn))
    _lib.check(rc, "ms_deform_attn_backward")
    return [grad_value, grad_loc, grad_attn]


# ---------------------------------------------------------------------------------------------------
# Fused prologue / epilogue (not part of the reference's pybind surface)
# ---------------------------------------------------------------------------------------------------
def fused_supported(value, reference_points, sampling_offsets, attn_logits):
    """The fused kernels cover the DINO configuration: fp32, 32 channels per head, reference dim 2 or 4."""
    return (value.is_cuda and value.dtype == torch.float32 and value.dim() == 4 and value.shape[3] == 32
            and reference_points.shape[-1] in (2, 4) and sampling_offsets.dtype == torch.float32
            and attn_logits.dtype == torch.float32 and reference_points.dtype == torch.float32)


def _fused_dims(value, spatial_shapes, level_start_index, reference_points, sampling_offsets, attn_logits):
    _check_inputs([("value", value), ("spatial_shapes", spatial_shapes), ("level_start_index", level_start_index),
                   ("reference_points", reference_points), ("sampling_offsets", sampling_offsets),
                   ("attention logits", attn_logits)])
    _assert(spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64,
            "expected scalar type Long for spatial_shapes / level_start_index")
    _assert(sampling_offsets.dim() == 6 and attn_logits.dim() == 4 and reference_points.dim() == 4,
            "ms_deform_attn_fused: expected sampling_offsets (N,Lq,M,L,P,2), logits (N,Lq,M,L*P), reference (N,Lq,L,2|4)")
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_offsets.shape
    _assert(tuple(attn_logits.shape) == (N, Lq, M, L * P) and tuple(reference_points.shape[:3]) == (N, Lq, L)
            and spatial_shapes.shape[0] == L, "ms_deform_attn_fused: inconsistent tensor shapes")
    if reference_points.shape[-1] not in (2, 4):
        raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead."
                         .format(reference_points.shape[-1]))
    return N, S, M, D, L, Lq, P


def ms_deform_attn_fused_forward(value, spatial_shapes, level_start_index, reference_points, sampling_offsets,
                                 attn_logits):
    """MSDeformAttn.forward between the Linear layers (ms_deform_attn.py:99-123) in one launch."""
    N, S, M, D, L, Lq, P = _fused_dims(value, spatial_shapes, level_start_index, reference_points,
                                       sampling_offsets, attn_logits)
    with torch.cuda.device(value.device):
        out = torch.empty((N, Lq, M * D), dtype=value.dtype, device=value.device)
        if out.numel() == 0 or value.numel() == 0:
            return out.zero_()
        rc = _lib.lib().semidetr_msda_fused_forward_f32(
            _lib.current_stream_ptr(), _p(value), _p(spatial_shapes), _p(level_start_index), _p(reference_points),
            reference_points.shape[-1], _p(sampling_offsets), _p(attn_logits), N, S, M, D, L, Lq, P, _p(out))
    _lib.check(rc, "ms_deform_attn_fused_forward")
    return out


def ms_deform_attn_fused_backward(value, spatial_shapes, level_start_index, reference_points, sampling_offsets,
                                  attn_logits, grad_output):
    """-> [grad_value, grad_sampling_offsets, grad_attn_logits]."""
    N, S, M, D, L, Lq, P = _fused_dims(value, spatial_shapes, level_start_index, reference_points,
                                       sampling_offsets, attn_logits)
    _assert(grad_output.is_contiguous() and grad_output.numel() == N * Lq * M * D and grad_output.dtype == value.dtype,
            "ms_deform_attn_fused_backward: grad_output must be a contiguous (N, Lq, M*D) tensor of value's dtype")
    with torch.cuda.device(value.device):
        grad_value = torch.empty_like(value)
        grad_off = torch.empty_like(sampling_offsets)
        grad_logit = torch.empty_like(attn_logits)
        if value.numel() == 0 or grad_off.numel() == 0:
            return [grad_value.zero_(), grad_off.zero_(), grad_logit.zero_()]
        rc = _lib.lib().semidetr_msda_fused_backward_f32(
            _lib.current_stream_ptr(), _p(grad_output), _p(value), _p(spatial_shapes), _p(level_start_index),
            _p(reference_points), reference_points.shape[-1], _p(sampling_offsets), _p(attn_logits), N, S, M, D, L,
            Lq, P, _p(grad_value), _p(grad_off), _p(grad_logit))
    _lib.check(rc, "ms_deform_attn_fused_backward")
    return [grad_value, grad_off, grad_logit]
