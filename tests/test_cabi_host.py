"""CPU: the C-ABI library loads and exports every symbol include/semidetr_hip.h declares, the ctypes
signature table covers them all, host-side argument checks work without a GPU, and the product fails
loudly (no fallback) when the library is missing or tensors are on the CPU."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header="semidetr_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(semidetr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes
    import semi_detr_amd
    names = _declared_functions()
    assert len(names) >= 12
    csrc = os.path.join(ROOT, "semi-detr_amd", "csrc")
    handle = ctypes.CDLL(os.path.join(csrc, "libsemidetr_hip.so"))
    for n in names:
        assert hasattr(handle, n), f"{n} declared in include/semidetr_hip.h but not exported"
    assert sorted(semi_detr_amd._lib.SIGNATURES) == names
    assert semi_detr_amd._lib.lib().semidetr_abi_version() == 7
    # the tuning / measurement entry points live ONLY in the experiments build (not in what ships)
    extra = _declared_functions("semidetr_hip_experiments.h")
    assert extra == sorted(semi_detr_amd._lib.EXPERIMENT_SIGNATURES) and len(extra) == 3
    exp = ctypes.CDLL(os.path.join(csrc, "libsemidetr_hip_exp.so"))
    for n in names + extra:
        assert hasattr(exp, n), f"{n} missing from libsemidetr_hip_exp.so"
    for n in extra:
        assert not hasattr(handle, n), f"{n} must not be exported by the product library"
    assert exp.semidetr_abi_version() == 7


def _code_object_kernels(path):
    """the AMDGPU code objects' kernel metadata of a host library: {demangled kernel name: metadata map} (msgpack notes,
    `amdhsa.kernels`; one note per translation unit)"""
    import subprocess

    import msgpack
    blob = open(path, "rb").read()
    out, at = {}, 0
    while True:
        at = blob.find(b"\xaeamdhsa.kernels", at)
        if at < 0:
            break
        u = msgpack.Unpacker(raw=False, strict_map_key=False)
        u.feed(blob[at - 1:at - 1 + (8 << 20)])                # the fixmap header precedes the first key
        for k in next(u)["amdhsa.kernels"]:
            out[k[".name"]] = k
        at += 1
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return {d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]: out[n] for n, d in zip(names, dem)}


SGPR_SPILL_CEILING_LOCATTN, SGPR_SPILL_CEILING_RAW = 53, 98      # measured maxima 45 / 90 (region scatter) + 8


def test_product_code_object_holds_only_reachable_msda_kernels():
    """the product library's MSDA code object = the kernels its dispatcher can reach (38: forward patch /
    strips x 3 splits, region-window forward for four and for five levels (round 4), generic, gather x 3, region scatter,
    1024-thread merged level scatter x 2, strips backward x 2 -- each for the reference contract and the fused prologue,
    + the two region-window instantiations that take the fused prologue's padding mask, the mask summary kernel and the
    lane-per-sample window gather x 3 (round 5),
    + fp64 generic), none of the rejected experiments -- of msda_rw_d32 only the forward configurations the dispatcher
    launches."""
    import subprocess
    csrc = os.path.join(ROOT, "semi-detr_amd", "csrc")
    syms = subprocess.run(["strings", "-a", os.path.join(csrc, "libsemidetr_hip.so")], capture_output=True, text=True).stdout
    kernels = set(re.findall(r"_ZN12_GLOBAL__N_1\d+(msda_[a-z0-9_]+)I[^\n]*?\.kd", syms))
    names = set(re.findall(r"(_ZN12_GLOBAL__N_1\d+msda_[A-Za-z0-9_]+)\.kd", syms))
    assert 20 <= len(names) <= 48, sorted(names)      # (round 6: + the five-level window gather x 3)
    for banned in ("msda_bwd_dest_d32", "msda_fwd_d32_lw", "msda_fwd_d32_res", "msda_bwd_enc_merged", "msda_bwd_encreg_merged",
                   "msda_bwd_lvl_coop", "msda_bwd_scatter_d32_win", "stream_kernel", "msda_bwd_own_merged",
                   "msda_fwd_d32_ws", "msda_bwd_lvl_mergedI", "msda_bwd_enc_fused_d32"):
        assert banned not in syms, banned
    # LocAttnIO + RawIO + RawIO-with-mask of <768, 25, 16, -1, 5, 4, forward> and <960, 24, 16, -1, 4, 5, forward>
    rw = sorted(n for n in names if "msda_rw_d32" in n)
    # (round 5: the four-level configuration twice -- with and without the tail split of small launches, msda_rw.h TUNE + 102400)
    assert len(rw) == 9 and sum("Li768ELi25ELi16ELin1ELi5ELi4ELb0E" in n for n in rw) == 6 and \
        sum("Li960ELi24ELi16ELin1ELi4ELi5ELb0E" in n for n in rw) == 3 and sum("ELb1EEEvPKf" in n for n in rw) == 3, rw      # (... MASK = true> of the fused prologue)
    # No kernel of the product spills VECTOR registers or uses SCRATCH (per-lane memory on gfx950: a handful of spilled registers
    # cost the window forward 30 % and hid the whole gain of the scatter's third workgroup per CU -- DESIGN.md section 6; a kernel can
    # also reach scratch with zero VGPR spills, when spilled SCALAR registers find no free lane: VERDICT r05, the five-level masked
    # window forward).  SCALAR registers spilled into lanes of a vector register the kernel holds (v_writelane / v_readlane, no
    # memory) are tolerated up to a ceiling per IO policy = the largest count measured when the ceiling was written + 8, so that a
    # change which makes them worse fails here with the kernel's NAME.
    meta = _code_object_kernels(os.path.join(csrc, "libsemidetr_hip.so"))
    msda = {n: k for n, k in meta.items() if n.startswith("msda_")}
    assert len(msda) >= 40, sorted(msda)
    bad = {n: (k[".vgpr_spill_count"], k[".private_segment_fixed_size"]) for n, k in meta.items()
           if k[".vgpr_spill_count"] or k[".private_segment_fixed_size"]}
    assert not bad, "VGPR spills / scratch: %r" % bad
    ceilings = {"LocAttnIO": SGPR_SPILL_CEILING_LOCATTN, "RawIO": SGPR_SPILL_CEILING_RAW}
    over = {n: k[".sgpr_spill_count"] for n, k in msda.items()
            if k[".sgpr_spill_count"] > next((c for io, c in ceilings.items() if io in n), 0)}
    assert not over, "SGPR spills above the policy's ceiling: %r" % over
    assert "getenv" not in subprocess.run(["nm", "-D", "--undefined-only", os.path.join(csrc, "libsemidetr_hip.so")],
                                          capture_output=True, text=True).stdout
    assert kernels >= {"msda_fwd_d32", "msda_rw_d32", "msda_bwd_gather_d32", "msda_bwd_scatter_d32_reg", "msda_bwd_lvl_merged_wide",
                       "msda_bwd_d32"}


def test_host_side_argument_errors_need_no_gpu():
    import semi_detr_amd
    lib = semi_detr_amd._lib.lib()
    rc = lib.semidetr_msda_forward_f32(None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 0, None)
    assert rc == -1 and b"null pointer" in lib.semidetr_last_error()
    rc = lib.semidetr_ema_flat_f32(None, None, None, -5, 0.5)
    assert rc == -1
    # round 4: the forward-kernel policy is validated on the host (0 adaptive, 1 patch, 2 window)
    assert lib.semidetr_msda_set_forward_policy(3) == -1 and b"0 adaptive" in lib.semidetr_last_error()
    assert lib.semidetr_msda_set_forward_policy(-1) == -1
    for ok in (2, 1, 0):
        assert lib.semidetr_msda_set_forward_policy(ok) == 0
    assert lib.semidetr_lsap_workspace_bytes(35, 900, 100) == 0          # fits LDS
    assert lib.semidetr_lsap_workspace_bytes(2, 22223, 10) > 0           # does not
    with pytest.raises(RuntimeError, match="code -1"):
        semi_detr_amd._lib.check(-1, "x")


def test_no_cpu_fallback_anywhere():
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as s
    v = torch.zeros(1, 4, 2, 2)
    sh = torch.tensor([[2, 2]])
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):     # ms_deform_attn.h:38
        MSDA.ms_deform_attn_forward(v, sh, torch.tensor([0]), torch.zeros(1, 1, 2, 1, 1, 2),
                                    torch.zeros(1, 1, 2, 1, 1), 64)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_backward(v, sh, torch.tensor([0]), torch.zeros(1, 1, 2, 1, 1, 2),
                                     torch.zeros(1, 1, 2, 1, 1), torch.zeros(1, 1, 4), 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.linear_sum_assignment(torch.zeros(3, 2))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        s.filter_pseudo_labels([torch.zeros(3, 5)], [torch.zeros(3)])
    with pytest.raises(RuntimeError):
        s.ema_update_([torch.zeros(3)], [torch.zeros(3)], 0.5)


def test_compiled_front_end_is_the_reference_module_surface():
    """`import MultiScaleDeformableAttention` resolves to the compiled extension's functions (the reference's pybind
    surface, src/vision.cpp:13-16); the cached pyramid check answers from the host copy of the level table."""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd
    assert type(MSDA.ms_deform_attn_forward).__name__ == "builtin_function_or_method" or "pybind" in repr(MSDA.ms_deform_attn_forward)
    assert semi_detr_amd.MultiScaleDeformableAttention._msda_ext.abi_version() == 7
    sh, ls = torch.tensor([[2, 3], [1, 2]]), torch.tensor([0, 6])
    assert MSDA.pyramid_check(sh, ls, 8) == 3
    assert MSDA.pyramid_check(sh, ls, 8) == 3                      # cache hit
    assert MSDA.pyramid_check(sh, torch.tensor([0, 5]), 8) == 1     # sum matches, levels overlap
    assert MSDA.pyramid_check(sh, ls, 9) == 0
    sh[1, 1] = 3                                                    # in-place edit bumps the version -> re-evaluated
    assert MSDA.pyramid_check(sh, ls, 8) == 0 and MSDA.pyramid_check(sh, ls, 9) == 3
    # round 5: the pixel-patch kernels pack a sample's top-left pixel into 15 + 15 bits -- a level side beyond 32766 is not vouched for
    assert MSDA.pyramid_check(torch.tensor([[1, 32766]]), torch.tensor([0]), 32766) == 3
    assert MSDA.pyramid_check(torch.tensor([[1, 32767]]), torch.tensor([0]), 32767) == 1
    with pytest.raises(RuntimeError, match="value tensor has to be contiguous|Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(torch.zeros(1, 4, 2, 2), sh, ls, torch.zeros(1, 1, 2, 2, 1, 2), torch.zeros(1, 1, 2, 2, 1), 64)


def test_pyramid_check_under_inference_mode_and_data_repointing():
    """inference tensors have no version counter (the cached check used to throw there, and with it the whole
    MSDA path under torch.inference_mode()); `t.data = ...` re-points a tensor without bumping its version."""
    import MultiScaleDeformableAttention as MSDA
    with torch.inference_mode():
        sh, ls = torch.tensor([[2, 3], [1, 2]]), torch.tensor([0, 6])
        assert MSDA.pyramid_check(sh, ls, 8) == 3
        assert MSDA.pyramid_check(sh, ls, 8) == 3
        sh[1, 1] = 3                                   # allowed on an inference tensor inside inference mode; never cached
        assert MSDA.pyramid_check(sh, ls, 8) == 0
    sh, ls = torch.tensor([[2, 3], [1, 2]]), torch.tensor([0, 6])
    assert MSDA.pyramid_check(sh, ls, 8) == 3
    sh.data = torch.tensor([[2, 3], [1, 3]])            # new storage, same version counter value
    assert MSDA.pyramid_check(sh, ls, 8) == 0


def test_missing_library_fails_loudly(monkeypatch):
    import semi_detr_amd
    monkeypatch.setattr(semi_detr_amd._lib, "_lib", None)
    monkeypatch.setattr(semi_detr_amd._lib, "LIB_PATH", "/nonexistent/libsemidetr_hip.so")
    with pytest.raises(semi_detr_amd._lib.NativeLibraryError, match="no CPU fallback"):
        semi_detr_amd._lib.lib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "semi-detr_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, re.M), f
                assert "liboracle" not in txt, f


def test_module_surface_matches_reference(golden_module):
    """Constructor kwargs, sub-module names / state_dict keys and the deterministic part of the init
    (ms_deform_attn.py:30-76) -- compared with the state_dict the reference's own module produced."""
    from semi_detr_amd import MSDeformAttn
    m = MSDeformAttn(d_model=32, n_levels=3, n_heads=4, n_points=2)
    want = {k[3:]: v for k, v in golden_module["ref2"].items() if k.startswith("sd.")}
    sd = m.state_dict()
    assert list(sd.keys()) == ["sampling_offsets.weight", "sampling_offsets.bias", "attention_weights.weight",
                               "attention_weights.bias", "value_proj.weight", "value_proj.bias",
                               "output_proj.weight", "output_proj.bias"]
    assert sorted(sd) == sorted(want)
    for k in sd:
        assert tuple(sd[k].shape) == want[k].shape
    np.testing.assert_array_equal(sd["sampling_offsets.bias"].numpy().astype(np.float64),
                                  want["sampling_offsets.bias"])          # direction grid, :62-70
    assert m.im2col_step == 64 and (m.d_model, m.n_levels, m.n_heads, m.n_points) == (32, 3, 4, 2)
    assert float(sd["value_proj.bias"].abs().max()) == 0 and float(sd["attention_weights.weight"].abs().max()) == 0
    with pytest.raises(ValueError, match="d_model must be divisible by n_heads"):
        MSDeformAttn(d_model=30, n_heads=8)


def test_assigner_surface_and_registry():
    import semi_detr_amd as s
    from semi_detr_amd import registry
    a = s.HungarianAssigner(cls_cost=dict(type="FocalLossCost", weight=2.0),
                            reg_cost=dict(type="BBoxL1Cost", weight=5.0, box_format="xywh"),
                            iou_cost=dict(type="IoUCost", iou_mode="giou", weight=2.0))
    assert (a.cls_cost.weight, a.cls_cost.alpha, a.cls_cost.gamma, a.cls_cost.eps) == (2.0, 0.25, 2, 1e-12)
    assert (a.reg_cost.weight, a.reg_cost.box_format) == (5.0, "xywh")
    assert (a.iou_cost.weight, a.iou_cost.iou_mode) == (2.0, "giou")
    with pytest.raises(AssertionError, match="gt_bboxes_ignore"):
        a.assign(torch.zeros(1, 4), torch.zeros(1, 2), torch.zeros(0, 4), torch.zeros(0), {}, gt_bboxes_ignore=1)
    with pytest.raises(KeyError):
        s.HungarianAssigner(cls_cost=dict(type="NoSuchCost"))
    done, skipped = registry.register_all()
    assert set(done) | set(skipped) == {"HungarianAssigner", "O2MAssigner", "BBoxL1Cost", "FocalLossCost", "IoUCost",
                                        "MeanTeacher", "TaskAlignedFocalLoss", "FlatDDP"}
    o = s.O2MAssigner()
    assert o.candidate_topk == 13 and s.O2MAssigner(candidate_topk=5).candidate_topk == 5
    with pytest.raises(AssertionError, match="gt_bboxes_ignore"):
        o.assign(torch.zeros(1, 4), torch.zeros(1, 2), torch.zeros(0, 4), torch.zeros(0), {}, gt_bboxes_ignore=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        o.assign(torch.zeros(20, 4), torch.zeros(20, 2), torch.zeros(1, 4), torch.zeros(1), dict(img_shape=(4, 4, 3)))
    h = s.MeanTeacher(momentum=0.999, interval=1, warm_up=0)
    for meth in ("before_run", "before_train_iter", "after_train_iter", "momentum_update"):
        assert callable(getattr(h, meth))
    assert s.ema_momentum(0.999, 0, 0) == 0.0          # step 0 copies the student (warm_up=0)


def test_graft_entry_build_runs():
    """The driver's build check: make (incremental) for the HIP library and the oracle, import, ABI version."""
    import __graft_entry__
    __graft_entry__.build()


def test_plugin_surface_builds_from_the_reference_config_dicts(monkeypatch):
    """`registry.register_all()` against registries that behave like mmcv's (register_module(name, force,
    module) / build(cfg)), pre-populated with placeholder classes under the reference's names (what importing detr_od /
    detr_ssod would have registered), then every object built from the LITERAL dicts of the reference's configs:
    configs/dino_detr/dino_detr_r50_8x2_12e_coco.py:40-44, configs/dino_detr/dino_detr_ssod_r50_coco_120k.py:30-34,45-51,
    configs/detr_ssod/detr_ssod_dino_detr_r50_coco_120k.py:43."""
    import sys
    import types

    class Registry:                       # the part of mmcv.utils.Registry the plugin surface uses
        def __init__(self, name):
            self.name, self.module_dict = name, {}

        def register_module(self, name=None, force=False, module=None):
            def _reg(cls):
                key = name or cls.__name__
                if not force and key in self.module_dict:
                    raise KeyError(f"{key} is already registered in {self.name}")
                self.module_dict[key] = cls
                return cls
            return _reg(module) if module is not None else _reg

        def build(self, cfg):
            args = dict(cfg)
            return self.module_dict[args.pop("type")](**args)

    regs = {k: Registry(k) for k in ("BBOX_ASSIGNERS", "MATCH_COST", "LOSSES", "HOOKS", "MODULE_WRAPPERS")}
    names = {"BBOX_ASSIGNERS": ["HungarianAssigner", "O2MAssigner"], "MATCH_COST": ["FocalLossCost", "BBoxL1Cost", "IoUCost"],
             "LOSSES": ["TaskAlignedFocalLoss"], "HOOKS": ["MeanTeacher"]}
    for reg, ns in names.items():         # the reference's own classes are registered first; force=True must replace them
        for n in ns:
            regs[reg].register_module(name=n, module=type(n, (), {"placeholder": True}))

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__path__ = []
        for k, v in attrs.items():
            setattr(m, k, v)
        monkeypatch.setitem(sys.modules, name, m)
    for pkg in ("mmdet", "mmdet.core", "mmdet.core.bbox", "mmdet.core.bbox.match_costs", "mmdet.models", "mmcv", "mmcv.runner"):
        mod(pkg)
    mod("mmdet.core.bbox.builder", BBOX_ASSIGNERS=regs["BBOX_ASSIGNERS"])
    mod("mmdet.core.bbox.match_costs.builder", MATCH_COST=regs["MATCH_COST"])
    mod("mmdet.models.builder", LOSSES=regs["LOSSES"])
    mod("mmcv.runner.hooks", HOOKS=regs["HOOKS"], Hook=object)
    mod("mmcv.parallel", MODULE_WRAPPERS=regs["MODULE_WRAPPERS"])

    import semi_detr_amd as s
    from semi_detr_amd import registry
    with pytest.raises(KeyError, match="already registered"):
        registry.register_all(force=False)
    done, skipped = registry.register_all()
    assert skipped == [] and sorted(done) == sorted(["HungarianAssigner", "O2MAssigner", "BBoxL1Cost", "FocalLossCost", "IoUCost",
                                                    "MeanTeacher", "TaskAlignedFocalLoss", "FlatDDP"])
    for reg in regs.values():
        assert not any(getattr(c, "placeholder", False) for c in reg.module_dict.values())

    # ---- the literal config dicts (copied as data)
    assigner = regs["BBOX_ASSIGNERS"].build(dict(
        type='HungarianAssigner',
        cls_cost=dict(type='FocalLossCost', weight=2.0),
        reg_cost=dict(type='BBoxL1Cost', weight=5.0, box_format='xywh'),
        iou_cost=dict(type='IoUCost', iou_mode='giou', weight=2.0)))
    assert isinstance(assigner, s.HungarianAssigner) and assigner.reg_cost.box_format == "xywh" and assigner.iou_cost.weight == 2.0
    assigner1 = regs["BBOX_ASSIGNERS"].build(dict(type='O2MAssigner'))
    assert isinstance(assigner1, s.O2MAssigner) and assigner1.candidate_topk == 13
    for cfg, attr in ((dict(type='FocalLossCost', weight=2.0), "alpha"), (dict(type='BBoxL1Cost', weight=5.0, box_format='xywh'),
                      "box_format"), (dict(type='IoUCost', iou_mode='giou', weight=2.0), "iou_mode")):
        c = regs["MATCH_COST"].build(cfg)
        assert c.weight == cfg["weight"] and hasattr(c, attr) and callable(c)
    loss = regs["LOSSES"].build(dict(type='TaskAlignedFocalLoss', use_sigmoid=True, gamma=2.0, loss_weight=2.0))
    assert isinstance(loss, torch.nn.Module) and (loss.gamma, loss.loss_weight, loss.use_sigmoid) == (2.0, 2.0, True)
    hook = regs["HOOKS"].build(dict(type="MeanTeacher", momentum=0.999, interval=1, warm_up=0))
    assert isinstance(hook, s.MeanTeacher) and (hook.momentum, hook.interval, hook.warm_up) == (0.999, 1, 0)
    from semi_detr_amd.dp import FlatDDP
    assert regs["MODULE_WRAPPERS"].module_dict["FlatDDP"] is FlatDDP
    # mmcv's is_module_wrapper(): isinstance(module, tuple(MODULE_WRAPPERS.module_dict.values()))
    wrapped = FlatDDP(torch.nn.Linear(2, 2))
    assert isinstance(wrapped, tuple(regs["MODULE_WRAPPERS"].module_dict.values())) and hasattr(wrapped, "module")
