"""ctypes binding of ``libsemidetr_hip.so`` (C ABI declared in ``include/semidetr_hip.h``).

There is NO fallback: if the shared library is missing or a call fails, an exception is raised.  The
library is built in-tree by ``__graft_entry__.build()`` / ``make -C semi-detr_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEMIDETR_EXPERIMENTS=1 (tests that force kernel variants, tools/, bench.py's HBM-peak probe) selects the experiments
# build: the same sources with every measured-and-rejected kernel variant + the tuning entry points
# (include/semidetr_hip_experiments.h).  The product library has neither.
EXPERIMENTS = os.environ.get("SEMIDETR_EXPERIMENTS", "0") not in ("", "0")
LIB_PATH = os.path.join(_HERE, "csrc", "libsemidetr_hip_exp.so" if EXPERIMENTS else "libsemidetr_hip.so")

c_void_p, c_int, c_int64, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double


class CostParams(ctypes.Structure):
    """Mirror of ``semidetr_cost_params`` (include/semidetr_hip.h)."""
    _fields_ = [("cls_weight", ctypes.c_float), ("alpha", ctypes.c_float), ("gamma", ctypes.c_float),
                ("eps", ctypes.c_float), ("reg_weight", ctypes.c_float), ("reg_xywh", ctypes.c_int),
                ("iou_weight", ctypes.c_float), ("iou_giou", ctypes.c_int), ("pred_xyxy", ctypes.c_int)]


_MSDA_FWD = [c_void_p] * 6 + [c_int] * 7 + [c_void_p]
_MSDA_BWD = [c_void_p] * 7 + [c_int] * 7 + [c_void_p] * 3
_MSDA_FWD_F32 = [c_void_p] * 6 + [c_int] * 8 + [c_void_p]           # f32 entry points carry `flags`
_MSDA_BWD_F32 = [c_void_p] * 7 + [c_int] * 8 + [c_void_p] * 3

# name -> (restype, argtypes); must list every function include/semidetr_hip.h declares
SIGNATURES = {
    "semidetr_abi_version": (c_int, []),
    "semidetr_last_error": (ctypes.c_char_p, []),
    "semidetr_msda_forward_f32": (c_int, _MSDA_FWD_F32),
    "semidetr_msda_forward_f64": (c_int, _MSDA_FWD),
    "semidetr_msda_backward_f32": (c_int, _MSDA_BWD_F32),
    "semidetr_msda_backward_f64": (c_int, _MSDA_BWD),
    "semidetr_msda_fused_forward_f32": (c_int, [c_void_p] * 5 + [c_int] + [c_void_p] * 4 + [c_int] * 8 + [c_void_p]),
    "semidetr_msda_fused_backward_f32": (c_int, [c_void_p] * 6 + [c_int] + [c_void_p] * 4 + [c_int] * 8 + [c_void_p] * 3),
    "semidetr_msda_mask_extents": (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_void_p]),
    "semidetr_msda_last_kernels": (ctypes.c_char_p, []),
    "semidetr_msda_set_forward_policy": (c_int, [c_int]),
    "semidetr_msda_forward_policy_state": (c_int, [c_void_p] * 4),
    "semidetr_msda_forward_policy_state_slot": (c_int, [c_int] + [c_void_p] * 4),
    "semidetr_msda_gather_choice": (c_int, [c_int]),
    "semidetr_match_cost_f32": (c_int, [c_void_p] * 7 + [c_int] * 4 + [ctypes.POINTER(CostParams), c_void_p]),
    "semidetr_lsap_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "semidetr_lsap_solve": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p] * 6),
    "semidetr_build_targets": (c_int, [c_void_p] * 6 + [c_int, c_int, c_int64] + [c_void_p] * 5),
    "semidetr_ema_multi_f32": (c_int, [c_void_p] * 5 + [c_int, c_int, c_double]),
    "semidetr_ema_flat_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_double]),
    "semidetr_pseudo_label_filter_f32": (c_int, [c_void_p] * 5 + [c_int] + [c_void_p] * 6),
    "semidetr_nms_workspace_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "semidetr_pseudo_nms_f32": (c_int, [c_void_p] * 4 + [c_int] * 3 + [ctypes.c_float, ctypes.c_float, c_int, c_void_p,
                                                                       ctypes.c_size_t] + [c_void_p] * 3),
    "semidetr_o2m_assign_f32": (c_int, [c_void_p] * 7 + [c_int] * 7 + [ctypes.c_float, ctypes.c_float] + [c_void_p] * 7),
    "semidetr_tal_loss_workspace_bytes": (ctypes.c_size_t, []),
    "semidetr_tal_loss_f32": (c_int, [c_void_p] * 4 + [c_int64, c_int, ctypes.c_float, c_int] + [c_void_p] * 3),
    "semidetr_transform_bboxes_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 3),
}

# include/semidetr_hip_experiments.h: only in libsemidetr_hip_exp.so
EXPERIMENT_SIGNATURES = {
    "semidetr_msda_set_variant": (None, [c_int, c_int]),
    "semidetr_debug_counters": (c_int, [c_void_p, c_int]),
    "semidetr_stream_copy_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def lib():
    """Load the shared library (once).  Raises ``NativeLibraryError`` when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C semi-detr_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        table = dict(SIGNATURES, **EXPERIMENT_SIGNATURES) if EXPERIMENTS else SIGNATURES
        for name, (res, args) in table.items():
            fn = getattr(handle, name)       # AttributeError if the .so is stale / symbol missing
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(rc, what):
    """Turn a C-ABI status into a RuntimeError (what the reference's AT_ASSERTM / launch failures give)."""
    if rc != 0:
        msg = lib().semidetr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def current_stream_ptr():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)


FORWARD_POLICIES = {"adaptive": 0, "patch": 1, "window": 2}


def set_forward_policy(policy):
    """Which kernel runs the encoder self-attention forward: "adaptive" (default: chosen from how far the samples of the
    previous launches reached), "patch" or "window" (include/semidetr_hip.h, semidetr_msda_set_forward_policy)."""
    check(lib().semidetr_msda_set_forward_policy(FORWARD_POLICIES.get(policy, policy)), "semidetr_msda_set_forward_policy")


def forward_policy_state(slot=0):
    """{'policy', 'mode' (0 patch / 1 window), 'far_fraction' (-1: no count received yet), 'updates'} of the current device and
    call-site slot (0 = the slot of callers that name none; MSDeformAttn instances hold theirs in ``policy_slot``)."""
    pol, mode, upd, frac = c_int(), c_int(), ctypes.c_uint(), ctypes.c_float()
    check(lib().semidetr_msda_forward_policy_state_slot(int(slot), ctypes.byref(pol), ctypes.byref(mode), ctypes.byref(frac),
                                                        ctypes.byref(upd)), "semidetr_msda_forward_policy_state_slot")
    return {"policy": pol.value, "mode": mode.value, "far_fraction": frac.value, "updates": upd.value}


def set_variant(fwd, bwd):
    """Force a kernel variant (experiments build only).  (0, 0) is always accepted: the product library has no variants."""
    if EXPERIMENTS:
        lib().semidetr_msda_set_variant(int(fwd), int(bwd))
    elif (fwd, bwd) != (0, 0):
        raise NativeLibraryError("kernel variants exist only in the experiments build: set SEMIDETR_EXPERIMENTS=1 "
                                 "(libsemidetr_hip_exp.so) before importing semi_detr_amd")
