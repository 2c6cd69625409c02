"""GPU: the lane-per-sample region-window GATHER of the encoder backward (csrc/msda_gw.h, round 5): grad_sampling_loc /
grad_attn_weight (and grad_value through the region scatter behind it) against the CPU oracle -- reference contract and fused
prologue / epilogue, with and without a padding mask, near samples (served from the LDS windows), far ones (the wave-cooperative
path) and samples outside the map; small pyramids incl. ragged regions and a non-halving one, and the full-size bs-4 launch.
Reference semantics: detr_od/models/utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-159, :301-403."""
import numpy as np
import pytest
import torch

import oracle
from conftest import kink_mask
from test_gpu_forward_policy import _band_mask, _case, _last
from test_gpu_fullsize import LEVELS, M, P, _encoder_case, _starts, _t
from test_gpu_fused import _prologue_np

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _window_policy():
    import semi_detr_amd as sda
    sda._lib.set_forward_policy("window")          # the backward's gather follows the slot's forward choice
    yield
    sda._lib.set_forward_policy("adaptive")


def _fused_expect(shp, loc, attn, o_gl, o_ga):
    """the fused epilogue restated on the oracle's gradients (ms_deform_attn.py:101-105 differentiated)"""
    N, Lq, M_, L, P_ = attn.shape
    norm = np.stack([shp[:, 1], shp[:, 0]], -1).astype(np.float64)[None, None, None, :, None, :]
    want_off = (o_gl.astype(np.float64) / norm).astype(np.float32)
    a64, g64 = attn.astype(np.float64).reshape(N, Lq, M_, L * P_), o_ga.astype(np.float64).reshape(N, Lq, M_, L * P_)
    return want_off, (a64 * (g64 - (a64 * g64).sum(-1, keepdims=True))).astype(np.float32)


@pytest.mark.parametrize("shapes,N,mode", [
    ([(20, 27), (10, 14), (5, 7), (3, 4)], 2, "near"),
    ([(20, 27), (10, 14), (5, 7), (3, 4)], 3, "far"),           # samples anywhere: (almost) every sample leaves its window
    ([(37, 53), (19, 27), (10, 14), (5, 7)], 2, "wide"),        # ragged regions, samples partly outside the map
    ([(16, 16), (16, 16), (15, 17), (2, 2)], 2, "near"),        # not a halving pyramid: any input is correct
    # five levels (round 6, the COCO-Full pyramid: 20 lanes per (query, head) row, three rows per wave, up-to-13 x 16 regions)
    ([(37, 53), (19, 27), (10, 14), (5, 7), (3, 4)], 2, "near"),
    ([(37, 53), (19, 27), (10, 14), (5, 7), (3, 4)], 2, "wide"),
    ([(20, 27), (10, 14), (5, 7), (3, 4), (2, 2)], 3, "far"),
    ([(16, 16), (16, 16), (15, 17), (2, 2), (1, 1)], 2, "near"),
])
def test_window_gather_reference_contract_vs_oracle(shapes, N, mode):
    import MultiScaleDeformableAttention as MSDA
    value, shp, loc, attn = _case(shapes, N, mode, 13)
    gout = np.random.default_rng(2).random((N, value.shape[1], M * 32)).astype(np.float32)
    o_gv, o_gl, o_ga = oracle.msda_backward(value, shp, loc, attn, gout)
    tsh = _t(shp)
    gv, gl, ga = MSDA.ms_deform_attn_backward(_t(value), tsh, _starts(tsh), _t(loc), _t(attn), _t(gout), 64)
    assert _last() == "msda_gw_d32+msda_bwd_scatter_d32_reg", _last()
    np.testing.assert_allclose(ga.cpu().numpy(), o_ga, rtol=0, atol=2e-5)
    np.testing.assert_allclose(gl.cpu().numpy(), o_gl, rtol=0, atol=1e-4 * max(1.0, float(np.abs(o_gl).max())))
    np.testing.assert_allclose(gv.cpu().numpy(), o_gv, rtol=0, atol=1e-4 * max(1.0, float(np.abs(o_gv).max())))


@pytest.mark.parametrize("shapes", [[(37, 53), (19, 27), (10, 14), (5, 7)], [(16, 16), (16, 16), (15, 17), (2, 2)],
                                    [(37, 53), (19, 27), (10, 14), (5, 7), (3, 4)]], ids=["pyramid", "no_pyramid", "five_levels"])
@pytest.mark.parametrize("kind", [None, "band", "band_with_holes", "random"])
@pytest.mark.parametrize("sigma", [2.0, 9.0, 40.0], ids=["near", "past_the_windows", "anywhere"])
def test_window_gather_fused_prologue_and_mask_vs_oracle(shapes, kind, sigma):
    """softmax + location arithmetic inside the kernel, the softmax backward through the DPP row sum, and the padding mask the
    reference always passes: summarised levels (band), byte-reading levels (holes / random); padded pixels hold NaN."""
    import MultiScaleDeformableAttention as MSDA
    N = 3
    value, shp, ref, off, logits, gout = _encoder_case(N, shapes, sigma, int(sigma) + 3)
    S = value.shape[1]
    mask = None
    if kind is not None:
        mask = _band_mask(shp, [(1.0, 1.0), (0.8, 0.55), (0.47, 0.93)])
        if kind == "band_with_holes":
            st = np.concatenate([[0], np.cumsum(shp[:, 0] * shp[:, 1])])
            mask[1, st[1] + 3] = True
            mask[1, st[len(shapes)] - 1] = False
        elif kind == "random":
            mask = np.random.default_rng(17).random((N, S)) < 0.15
    loc, attn = _prologue_np(ref, off, logits, shp, P)
    vm = value.copy()
    if mask is not None:
        vm[mask] = 0
    o_gv, o_gl, o_ga = oracle.msda_backward(vm, shp, loc, attn, gout)
    if mask is not None:
        o_gv[mask] = 0
        value[mask] = np.nan
    want_off, want_log = _fused_expect(shp, loc, attn, o_gl, o_ga)
    tsh = _t(shp)
    gv, goff, glog = MSDA.ms_deform_attn_fused_backward(_t(value), tsh, _starts(tsh), _t(ref), _t(off), _t(logits), _t(gout),
                                                        None if mask is None else _t(mask))
    assert _last() == "msda_gw_d32+msda_bwd_scatter_d32_reg", _last()
    ok = ~kink_mask(loc, shp)
    np.testing.assert_allclose(glog.cpu().numpy(), want_log, rtol=0, atol=2e-5)
    np.testing.assert_allclose(goff.cpu().numpy()[ok], want_off[ok], rtol=0, atol=1e-4 * max(1.0, float(np.abs(want_off).max())))
    gv = gv.cpu().numpy()
    if mask is not None:
        assert np.all(gv[mask] == 0.0)
    np.testing.assert_allclose(gv, o_gv, rtol=0, atol=1e-4 * max(1.0, float(np.abs(o_gv).max())))


@pytest.mark.parametrize("levels", [LEVELS, LEVELS + [(7, 11)]], ids=["four_levels", "five_levels"])
@pytest.mark.parametrize("io", ["locattn", "raw", "raw_masked"])
def test_window_gather_full_size_vs_oracle(io, levels):
    """N = 4, Lq = S = 22 223 (22 300 with the COCO-Full recipe's fifth level), sigma 2 px (the launches bench.py times): every
    element of the two small gradients."""
    import MultiScaleDeformableAttention as MSDA
    value, shp, ref, off, logits, gout = _encoder_case(4, levels, 2.0, 29)
    mask = _band_mask(shp, [(1.0, 1.0), (0.85, 0.6), (0.6, 1.0), (0.75, 0.75)]) if io == "raw_masked" else None
    loc, attn = _prologue_np(ref, off, logits, shp, P)
    vm = value.copy()
    if mask is not None:
        vm[mask] = 0
    _, o_gl, o_ga = oracle.msda_backward(vm, shp, loc, attn, gout)
    tsh = _t(shp)
    tls = _starts(tsh)
    if io == "locattn":
        _, gl, ga = MSDA.ms_deform_attn_backward(_t(value), tsh, tls, _t(loc), _t(attn), _t(gout), 64)
        assert _last() == "msda_gw_d32+msda_bwd_scatter_d32_reg", _last()
        np.testing.assert_allclose(ga.cpu().numpy(), o_ga, rtol=0, atol=2e-5)
        np.testing.assert_allclose(gl.cpu().numpy(), o_gl, rtol=0, atol=1e-4 * max(1.0, float(np.abs(o_gl).max())))
        return
    want_off, want_log = _fused_expect(shp, loc, attn, o_gl, o_ga)
    _, goff, glog = MSDA.ms_deform_attn_fused_backward(_t(value), tsh, tls, _t(ref), _t(off), _t(logits), _t(gout),
                                                       None if mask is None else _t(mask))
    assert _last() == "msda_gw_d32+msda_bwd_scatter_d32_reg", _last()
    ok = ~kink_mask(loc, shp)
    np.testing.assert_allclose(glog.cpu().numpy(), want_log, rtol=0, atol=2e-5)
    np.testing.assert_allclose(goff.cpu().numpy()[ok], want_off[ok], rtol=0, atol=1e-4 * max(1.0, float(np.abs(want_off).max())))
