"""``MultiScaleDeformableAttention`` -- the native module the reference imports
(``import MultiScaleDeformableAttention as MSDA``, detr_od/models/utils/ops/functions/ms_deform_attn_func.py:18).

The reference builds it with pybind from ``ops/src`` (vision.cpp:13-16).  Here it is the compiled extension
``_msda_ext`` (``csrc/msda_ext.cpp``: at::Tensor signatures of src/ms_deform_attn.h:20-61, same argument order, same
precondition errors, outputs freshly allocated on the inputs' device, current stream, no synchronisation) in front of
the C ABI of ``csrc/libsemidetr_hip.so`` where the hand-written gfx950 kernels live.  This file only loads it and
re-exports its functions; there is no Python or CPU implementation to fall back to.

    ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step) -> Tensor
    ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output,
                            im2col_step) -> [grad_value, grad_sampling_loc, grad_attn_weight]
    ms_deform_attn_fused_forward / ms_deform_attn_fused_backward / fused_supported   
    pyramid_check(spatial_shapes, level_start_index, S) -> bit 0: sum(H*W) == S, bit 1: exact tiling (cached)
    mask_extents(padding_mask, spatial_shapes, level_start_index) -> (N, L) int32 summary of a padding mask (cached; the
                                                                      fused calls fetch it themselves)
"""
import torch  # noqa: F401  (libtorch must be loaded before the extension)

from . import _lib

try:
    if _lib.EXPERIMENTS:      # the front end linked against libsemidetr_hip_exp.so
        from . import _msda_ext_exp as _msda_ext
    else:
        from . import _msda_ext
except ImportError as e:      # not built (or built against another torch): fail loudly, never degrade
    raise _lib.NativeLibraryError(
        "semi-detr_amd/_msda_ext*.so is missing or does not load (%s): build it with "
        "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C semi-detr_amd/csrc`. "
        "There is no CPU fallback." % (e,)) from e

ms_deform_attn_forward = _msda_ext.ms_deform_attn_forward
ms_deform_attn_backward = _msda_ext.ms_deform_attn_backward
ms_deform_attn_fused_forward = _msda_ext.ms_deform_attn_fused_forward
ms_deform_attn_fused_backward = _msda_ext.ms_deform_attn_fused_backward
fused_supported = _msda_ext.fused_supported
pyramid_check = _msda_ext.pyramid_check
mask_extents = _msda_ext.mask_extents
gather_choice = _msda_ext.gather_choice

if _msda_ext.abi_version() != _lib.lib().semidetr_abi_version():      # a stale front end against a newer library
    raise _lib.NativeLibraryError("semi-detr_amd/_msda_ext*.so was built against ABI %d, libsemidetr_hip.so is ABI %d: rebuild "
                                  "(make -C semi-detr_amd/csrc)" % (_msda_ext.abi_version(), _lib.lib().semidetr_abi_version()))
