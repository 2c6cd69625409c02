"""GPU: the two kernels of the encoder self-attention FORWARD on the PRODUCT library and the data-driven choice between them
(include/semidetr_hip.h: semidetr_msda_set_forward_policy).

  * parity: patch kernel (policy "patch") and region-window kernel (policy "window") against the CPU oracle on the same
    inputs -- small pyramids incl. ragged edges / far samples / samples outside the map, the full-size bs-4 encoder shape
    for the reference contract AND the fused prologue (four levels and the five-level COCO-Full pyramid), and the patch
    kernel's results where the window kernel does not apply (padding mask, six levels);
  * the adaptive policy: close samples move the dispatcher to the window kernel, far ones move it back, and what the
    library reports as launched is what the policy state says.
"""
import numpy as np
import pytest
import torch

import oracle
from test_gpu_fullsize import LEVELS, M, D, P, _encoder_case, _starts, _t
from test_gpu_fused import _prologue_np

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _restore_policy():
    import semi_detr_amd as sda
    yield
    sda._lib.set_forward_policy("adaptive")


def _last():
    import semi_detr_amd as sda
    return sda._lib.lib().semidetr_msda_last_kernels().decode()


def _case(shapes, N, mode, seed):
    rng = np.random.default_rng(seed)
    shp = np.asarray(shapes, np.int64)
    L = len(shapes)
    S = int((shp[:, 0] * shp[:, 1]).sum())
    ref = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2)
                          for h, w in shapes])
    if mode == "near":
        loc = ref[None, :, None, None, None, :] + rng.standard_normal((N, S, M, L, P, 2)) * (2.0 / shp[None, None, None, :, None, ::-1])
    elif mode == "far":
        loc = rng.random((N, S, M, L, P, 2))
    else:                                   # wide: partly outside the map
        loc = ref[None, :, None, None, None, :] + rng.standard_normal((N, S, M, L, P, 2)) * 0.3
    value = (rng.random((N, S, M, D)) * 0.01).astype(np.float32)
    attn = rng.random((N, S, M, L, P)) + 1e-5
    attn /= attn.sum((-1, -2), keepdims=True)
    return value, shp, loc.astype(np.float32), attn.astype(np.float32)


@pytest.mark.parametrize("shapes,N,mode", [
    ([(20, 27), (10, 14), (5, 7), (3, 4)], 2, "near"),          # DINO-like pyramid
    ([(20, 27), (10, 14), (5, 7), (3, 4)], 3, "far"),           # samples anywhere: (almost) every sample leaves its window
    ([(37, 53), (19, 27), (10, 14), (5, 7)], 2, "wide"),        # ragged 16 x 16 regions, samples partly outside the map
    ([(16, 16), (16, 16), (15, 17), (2, 2)], 2, "near"),        # levels that are NOT a halving pyramid: any input is correct
    ([(37, 53), (19, 27), (10, 14), (5, 7), (3, 4)], 2, "near"),    # five levels (COCO-Full pyramid): its own instantiation
    ([(37, 53), (19, 27), (10, 14), (5, 7), (3, 4)], 2, "wide"),
    ([(20, 27), (10, 14), (5, 7), (3, 4), (2, 2)], 3, "far"),
])
def test_window_and_patch_forward_vs_oracle(shapes, N, mode):
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    value, shp, loc, attn = _case(shapes, N, mode, 5)
    want = oracle.msda_forward(value, shp, loc, attn)
    tsh = _t(shp)
    args = (_t(value), tsh, _starts(tsh), _t(loc), _t(attn), 64)
    for policy, kernel in (("patch", "msda_fwd_d32<1, 4, 408"), ("window", "msda_rw_d32")):
        sda._lib.set_forward_policy(policy)
        out = MSDA.ms_deform_attn_forward(*args)
        assert _last() == kernel, (policy, _last())
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-6, err_msg=policy)


@pytest.mark.parametrize("io", ["locattn", "raw"])
@pytest.mark.parametrize("levels", [LEVELS, LEVELS + [(7, 11)]], ids=["four_levels", "five_levels"])
def test_window_forward_full_size_vs_oracle(io, levels):
    """N = 4, Lq = S = 22 223 (the launch bench.py times) and the five-level COCO-Full pyramid (S = 22 300), sigma 2 px: the
    window kernel against the oracle, every element, for the reference contract and for the fused prologue (softmax +
    locations inside the kernel)."""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    value, shp, ref, off, logits, _ = _encoder_case(4, levels, 2.0, 21)
    loc, attn = _prologue_np(ref, off, logits, shp, P)
    want = oracle.msda_forward(value, shp, loc, attn)
    tsh = _t(shp)
    sda._lib.set_forward_policy("window")
    if io == "locattn":
        out = MSDA.ms_deform_attn_forward(_t(value), tsh, _starts(tsh), _t(loc), _t(attn), 64)
    else:
        out = MSDA.ms_deform_attn_fused_forward(_t(value), tsh, _starts(tsh), _t(ref), _t(off), _t(logits))
    assert _last() == "msda_rw_d32"
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-6)


def test_window_policy_keeps_the_patch_kernel_where_the_window_kernel_does_not_apply():
    """Six levels, a padding mask: policy "window" must fall back to the patch kernel (and give its results)."""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    sda._lib.set_forward_policy("window")
    value, shp, loc, attn = _case([(20, 27), (10, 14), (5, 7), (3, 4)], 1, "near", 8)      # ONE image does take the window kernel
    tsh = _t(shp)
    out = MSDA.ms_deform_attn_forward(_t(value), tsh, _starts(tsh), _t(loc), _t(attn), 64)
    assert _last() == "msda_rw_d32", _last()
    np.testing.assert_allclose(out.cpu().numpy(), oracle.msda_forward(value, shp, loc, attn), rtol=0, atol=2e-6)
    for shapes, N in (([(20, 27), (10, 14), (5, 7), (3, 4), (2, 2), (1, 1)], 2),):
        value, shp, loc, attn = _case(shapes, N, "near", 9)
        tsh = _t(shp)
        out = MSDA.ms_deform_attn_forward(_t(value), tsh, _starts(tsh), _t(loc), _t(attn), 64)
        assert _last() == "msda_fwd_d32<1, 4, 408", (N, len(shapes), _last())
        np.testing.assert_allclose(out.cpu().numpy(), oracle.msda_forward(value, shp, loc, attn), rtol=0, atol=2e-6)
    # fused prologue with a padding mask
    value, shp, ref, off, logits, _ = _encoder_case(2, [(20, 27), (10, 14), (5, 7), (3, 4)], 2.0, 3)
    S = value.shape[1]
    mask = np.zeros((2, S), np.uint8)
    mask[1, ::7] = 1
    tsh = _t(shp)
    out = MSDA.ms_deform_attn_fused_forward(_t(value), tsh, _starts(tsh), _t(ref), _t(off), _t(logits), _t(mask).bool())
    assert _last() == "msda_fwd_d32<1, 4, 408"
    loc, attn = _prologue_np(ref, off, logits, shp, P)
    vm = value.copy()
    vm[mask.astype(bool)] = 0
    np.testing.assert_allclose(out.cpu().numpy(), oracle.msda_forward(vm, shp, loc, attn), rtol=0, atol=2e-6)


def test_adaptive_policy_follows_the_sample_spread():
    """Close samples (sigma 1 px) -> the dispatcher moves to the window kernel; far samples (sigma 7 px) -> back to the patch
    kernel; results equal the oracle's throughout.  (The count of launch k reaches the host when launch k + 1 starts and is
    acted on by the dispatch after that, so a few launches with a synchronisation in between are needed: in training the
    host runs ahead and the choice simply lags by a step.)"""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    sda._lib.set_forward_policy("adaptive")
    shapes = [(40, 54), (20, 27), (10, 14), (5, 7)]
    shp = np.asarray(shapes, np.int64)
    tsh = _t(shp)
    tls = _starts(tsh)

    def run(sigma, launches):
        value, _, ref, off, logits, _ = _encoder_case(2, shapes, sigma, int(sigma * 10))
        loc, attn = _prologue_np(ref, off, logits, shp, P)
        want = oracle.msda_forward(value, shp, loc, attn)
        a = (_t(value), tsh, tls, _t(loc), _t(attn), 64)
        kernels = []
        for _ in range(launches):
            out = MSDA.ms_deform_attn_forward(*a)
            kernels.append(_last())
            torch.cuda.synchronize()
            np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-6)
        return kernels

    st0 = sda._lib.forward_policy_state()
    k_close = run(1.0, 6)
    st1 = sda._lib.forward_policy_state()
    assert st1["updates"] > st0["updates"], (st0, st1)
    assert st1["mode"] == 1 and 0.0 <= st1["far_fraction"] < 0.60, st1
    assert k_close[-1] == "msda_rw_d32", k_close
    k_far = run(7.0, 6)
    st2 = sda._lib.forward_policy_state()
    assert st2["mode"] == 0 and st2["far_fraction"] > 0.70, st2
    assert k_far[0] == "msda_rw_d32" and k_far[-1] == "msda_fwd_d32<1, 4, 408", k_far


@pytest.mark.parametrize("policy", ["window", "adaptive"])
def test_forward_inside_a_stream_capture(policy):
    """The encoder forward captured into a HIP graph and replayed: the dispatcher must not allocate, copy or synchronise while
    the stream is capturing (include/semidetr_hip.h: "launches inside a stream capture keep the kernel of the moment and count
    nothing"), and the replayed launch gives the oracle's result for new contents of the same buffers."""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    shapes = [(40, 54), (20, 27), (10, 14), (5, 7)]
    value, shp, loc, attn = _case(shapes, 2, "near", 31)
    tsh = _t(shp)
    tls = _starts(tsh)
    tv, tl, ta = _t(value), _t(loc), _t(attn)
    sda._lib.set_forward_policy(policy)
    for _ in range(3):                                     # warm-up outside the capture (first-use allocations, LDS attribute)
        MSDA.ms_deform_attn_forward(tv, tsh, tls, tl, ta, 64)
        torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = MSDA.ms_deform_attn_forward(tv, tsh, tls, tl, ta, 64)
    torch.cuda.current_stream().wait_stream(s)
    before = sda._lib.forward_policy_state()["updates"]      # (the capturing dispatch may still have picked up the warm-up launches' counts)
    value2, _, loc2, attn2 = _case(shapes, 2, "near", 32)
    tv.copy_(_t(value2)); tl.copy_(_t(loc2)); ta.copy_(_t(attn2))
    g.replay()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), oracle.msda_forward(value2, shp, loc2, attn2), rtol=0, atol=2e-6)
    assert sda._lib.forward_policy_state()["updates"] == before      # the replayed launch counted nothing
