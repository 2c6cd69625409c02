"""GPU: the two kernels of the encoder self-attention FORWARD on the PRODUCT library and the data-driven choice between them
(include/semidetr_hip.h: semidetr_msda_set_forward_policy).

  * parity: patch kernel (policy "patch") and region-window kernel (policy "window") against the CPU oracle on the same
    inputs -- small pyramids incl. ragged edges / far samples / samples outside the map, the full-size bs-4 encoder shape
    for the reference contract AND the fused prologue (four levels and the five-level COCO-Full pyramid), and the patch
    kernel's results where the window kernel does not apply (six levels); the fused prologue WITH a padding mask -- what the
    reference's encoder always passes -- through both kernels: small pyramids with band / random / whole-image padding at three
    sample spreads, and four images of different sizes on the full-size canvas;
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


@pytest.mark.parametrize("io", ["locattn", "raw"])
@pytest.mark.parametrize("N", [1, 2])
def test_window_forward_tail_split_full_size_vs_oracle(io, N):
    """Round 5: launches of fewer than ~three waves of workgroups run the instantiation with the TAIL SPLIT (msda_rw.h, TUNE + 102400) --
    one image is 352 (image, head, region) units on 256 CUs: the last 96 are cut into two parts of their rounds, the second parts
    taken by helper workgroups; two images are 704 units, 192 in the tail.  Every element against the oracle."""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    value, shp, ref, off, logits, _ = _encoder_case(N, LEVELS, 2.0, 31 + N)
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
    """Six levels: policy "window" must fall back to the patch kernel (and give its results)."""
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


def _band_mask(shp, fracs):
    """Padding as the reference builds it (detr_od/models/dense_heads/dino_detr_head.py:305-318: images smaller than the batch
    canvas): on every level the bottom / right band beyond (fh * H, fw * W) of image n is padding.  (N, S) bool."""
    rows = []
    for fh, fw in fracs:
        per = []
        for h, w in shp:
            mk = np.zeros((int(h), int(w)), bool)
            mk[int(np.ceil(fh * h)):, :] = True
            mk[:, int(np.ceil(fw * w)):] = True
            per.append(mk.reshape(-1))
        rows.append(np.concatenate(per))
    return np.stack(rows)


def test_mask_extents_summary():
    """semidetr_msda_mask_extents: vh | vw << 16 exactly when a level's padding is "rows >= vh or columns >= vw", -1 otherwise;
    the cached answer follows in-place changes of the mask (version counter)."""
    import MultiScaleDeformableAttention as MSDA
    shapes = [(37, 53), (19, 27), (10, 14), (5, 7), (1, 1)]
    shp = np.asarray(shapes, np.int64)
    st = np.concatenate([[0], np.cumsum(shp[:, 0] * shp[:, 1])])
    mask = _band_mask(shp, [(1.0, 1.0), (0.8, 0.55), (0.0, 0.3), (0.5, 1.0), (1.0, 0.5)])
    mask[3, st[1] + 2 * 27 + 5] = True              # image 3, level 1: a padded pixel inside the valid area
    mask[4, st[3] - 1] = False                      # image 4, level 2: a valid pixel inside the band
    want = []
    for fh, fw in [(1.0, 1.0), (0.8, 0.55), (0.0, 0.3), (0.5, 1.0), (1.0, 0.5)]:
        want.append([[int(np.ceil(fh * h)), int(np.ceil(fw * w))] for h, w in shapes])
    want = np.asarray(want)
    want[2] = 0                                     # nothing valid: {0, 0} whatever the other extent says
    want = want[..., 0] | (want[..., 1] << 16)
    want[3, 1] = -1
    want[4, 2] = -1
    tsh, tm = _t(shp), _t(mask)
    tls = _starts(tsh)
    got = MSDA.mask_extents(tm, tsh, tls)
    assert got.dtype == torch.int32 and tuple(got.shape) == (5, 5)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    assert MSDA.mask_extents(tm, tsh, tls).data_ptr() == got.data_ptr()          # cached
    tm[3, int(st[1]) + 2 * 27 + 5] = False                                        # in place: the version moves, the summary follows
    want[3, 1] = int(np.ceil(0.5 * 19)) | (27 << 16)
    np.testing.assert_array_equal(MSDA.mask_extents(tm, tsh, tls).cpu().numpy(), want)
    np.testing.assert_array_equal(MSDA.mask_extents(tm.to(torch.uint8), tsh, tls).cpu().numpy(), want)
    rnd = _t(np.random.default_rng(0).random((2, int(st[-1]))) < 0.5)
    assert (MSDA.mask_extents(rnd, tsh, tls)[:, :4] == -1).all()


def _masked_fused_forward(value, shp, ref, off, logits, mask):
    import MultiScaleDeformableAttention as MSDA
    tsh = _t(shp)
    return MSDA.ms_deform_attn_fused_forward(_t(value), tsh, _starts(tsh), _t(ref), _t(off), _t(logits), _t(mask))


@pytest.mark.parametrize("shapes", [[(37, 53), (19, 27), (10, 14), (5, 7)], [(37, 53), (19, 27), (10, 14), (5, 7), (3, 4)],
                                    [(16, 16), (16, 16), (15, 17), (2, 2)]], ids=["four_levels", "five_levels", "no_pyramid"])
@pytest.mark.parametrize("kind", ["band", "band_with_holes", "random", "one_image_all_padding", "no_padding"])
@pytest.mark.parametrize("sigma", [2.0, 9.0, 40.0], ids=["near", "past_the_mask_window", "anywhere"])
def test_window_forward_with_padding_mask_vs_oracle(shapes, kind, sigma):
    """The reference ALWAYS hands MSDeformAttn a padding mask (transformer.py:1309,1460), so the fused prologue's default path
    must reach the region-window kernel with one: staged rows of padded pixels are zeros, level-0 corners come
    from the LDS mask window (sigma 2 px), from global mask bytes (9 px: beyond the window's 8-pixel margin) and through the
    out-of-window path of the coarse levels (40 px) -- all equal to the oracle on `value.masked_fill(mask, 0)`
    (ops/modules/ms_deform_attn.py:95-96).  Padded pixels hold NaN: they must never reach a result."""
    import semi_detr_amd as sda
    N = 3
    value, shp, ref, off, logits, _ = _encoder_case(N, shapes, sigma, int(sigma) + len(shapes))
    S = value.shape[1]
    rng = np.random.default_rng(17)
    if kind == "band":                  # every level summarised as {vh, vw}: corners are tested with two compares
        mask = _band_mask(shp, [(1.0, 1.0), (0.8, 0.55), (0.47, 0.93)])
    elif kind == "band_with_holes":     # levels 1 and 3 of image 1 are NOT a band (one more padded / one valid pixel): those two read bytes
        mask = _band_mask(shp, [(1.0, 1.0), (0.8, 0.55), (0.47, 0.93)])
        st = np.concatenate([[0], np.cumsum(shp[:, 0] * shp[:, 1])])
        mask[1, st[1] + 3] = True
        mask[1, st[4] - 1] = False
    elif kind == "random":              # no level has the form: every corner reads its byte
        mask = rng.random((N, S)) < 0.15
    elif kind == "one_image_all_padding":
        mask = np.zeros((N, S), bool)
        mask[1] = True
    else:
        mask = np.zeros((N, S), bool)
    loc, attn = _prologue_np(ref, off, logits, shp, P)
    vm = value.copy()
    vm[mask] = 0
    want = oracle.msda_forward(vm, shp, loc, attn)
    value[mask] = np.nan
    for policy, kernel in (("window", "msda_rw_d32"), ("patch", "msda_fwd_d32<1, 4, 408")):
        sda._lib.set_forward_policy(policy)
        out = _masked_fused_forward(value, shp, ref, off, logits, mask)
        assert _last() == kernel, (policy, _last())
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-6, err_msg=policy)


@pytest.mark.parametrize("levels", [LEVELS, LEVELS + [(7, 11)]], ids=["four_levels", "five_levels"])
def test_window_forward_full_size_mixed_image_shapes_vs_oracle(levels):
    """N = 4 images of DIFFERENT sizes on one 800 x 1333 canvas (the step's encoder call: `mask_flatten` of a padded batch),
    Lq = S, sigma 2 px: fused prologue + mask through the window kernel, every element against the oracle; and the backward of
    the same call (patch gather + region scatter) leaves exactly zero gradient on the padded rows."""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    value, shp, ref, off, logits, gout = _encoder_case(4, levels, 2.0, 23)
    mask = _band_mask(shp, [(1.0, 1.0), (0.85, 0.6), (0.6, 1.0), (0.75, 0.75)])
    assert 0.2 < mask.mean() < 0.4
    loc, attn = _prologue_np(ref, off, logits, shp, P)
    vm = value.copy()
    vm[mask] = 0
    want = oracle.msda_forward(vm, shp, loc, attn)
    o_gv, _, _ = oracle.msda_backward(vm, shp, loc, attn, gout)
    o_gv[mask] = 0                                                # masked_fill's backward
    value[mask] = np.nan
    sda._lib.set_forward_policy("window")
    out = _masked_fused_forward(value, shp, ref, off, logits, mask)
    assert _last() == "msda_rw_d32"
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-6)
    tsh = _t(shp)
    gv, _, _ = MSDA.ms_deform_attn_fused_backward(_t(value), tsh, _starts(tsh), _t(ref), _t(off), _t(logits), _t(gout), _t(mask))
    gv = gv.cpu().numpy()
    assert np.all(gv[mask] == 0.0), "padded pixels must receive exactly zero gradient"
    np.testing.assert_allclose(gv, o_gv, rtol=0, atol=1e-4 * max(1.0, float(np.abs(o_gv).max())))


def test_adaptive_policy_follows_the_sample_spread():
    """Close samples (sigma 1 px) -> the dispatcher moves to the window kernel; far samples (sigma 14 px) -> back to the patch
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
    assert st1["mode"] == 1 and 0.0 <= st1["far_fraction"] < 0.72, st1
    assert k_close[-1] == "msda_rw_d32", k_close
    k_far = run(14.0, 6)
    st2 = sda._lib.forward_policy_state()
    assert st2["mode"] == 0 and st2["far_fraction"] > 0.80, st2
    assert k_far[0] == "msda_rw_d32" and k_far[-1] == "msda_fwd_d32<1, 4, 408", k_far


def test_adaptive_policy_is_kept_per_call_site():
    """Two "layers" whose offsets reach differently far (sigma 1 px and 14 px), called alternately as the layers of an encoder are
    (the reference builds twelve MSDeformAttn instances per model, transformer.py:609,760): with a slot each
    (SEMIDETR_MSDA_POLICY_SLOT, the module's `policy_slot`) each settles on ITS kernel and stays there, whatever the other one
    does; the shared slot 0 is not touched.  Results equal the oracle's throughout."""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    sda._lib.set_forward_policy("adaptive")
    shapes = [(40, 54), (20, 27), (10, 14), (5, 7)]
    shp = np.asarray(shapes, np.int64)
    tsh = _t(shp)
    tls = _starts(tsh)
    layers = {}
    for slot, sigma in ((7, 1.0), (8, 14.0)):
        value, _, ref, off, logits, _ = _encoder_case(2, shapes, sigma, 40 + slot)
        loc, attn = _prologue_np(ref, off, logits, shp, P)
        layers[slot] = ((_t(value), tsh, tls, _t(loc), _t(attn), 64, slot), oracle.msda_forward(value, shp, loc, attn))
    before0 = sda._lib.forward_policy_state(0)["updates"]
    seen = {7: [], 8: []}
    for it in range(10):
        for slot, (a, want) in layers.items():
            out = MSDA.ms_deform_attn_forward(*a)
            seen[slot].append(_last())
            torch.cuda.synchronize()
            np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=2e-6)
    st7, st8 = sda._lib.forward_policy_state(7), sda._lib.forward_policy_state(8)
    assert st7["mode"] == 1 and st7["far_fraction"] < 0.72 and st7["updates"] >= 5, st7
    assert st8["mode"] == 0 and st8["far_fraction"] > 0.80 and st8["updates"] >= 5, st8
    assert set(seen[7][4:]) == {"msda_rw_d32"}, seen[7]                 # settled after a few launches, then never moved
    assert set(seen[8]) == {"msda_fwd_d32<1, 4, 408"}, seen[8]
    assert sda._lib.forward_policy_state(0)["updates"] == before0
    # the nn.Module: every instance takes its own slot
    from semi_detr_amd import MSDeformAttn
    a, b = MSDeformAttn(), MSDeformAttn()
    assert 1 <= a.policy_slot <= 255 and b.policy_slot == a.policy_slot % 255 + 1


def test_module_steps_reach_the_window_kernels_under_the_default_policy():
    """What a checkout with unchanged configs gets: two `MSDeformAttn` layers with the reference's call structure --
    padding mask of a batch of differently sized images, valid-ratio scaled reference points -- stepped a few times under the
    DEFAULT (adaptive) policy.  Each instance's own slot sees near samples, so after the counts of the first iterations have arrived
    the forward is the region-window kernel and the backward's gather the lane-per-sample window gather; loss and parameter
    gradients equal the patch kernels' (policy "patch") to fp32 rounding."""
    import bench
    import semi_detr_amd as sda
    from semi_detr_amd import MSDeformAttn
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    layers = [MSDeformAttn(256, 4, 8, 4).to(dev) for _ in range(2)]
    for m in layers:
        with torch.no_grad():
            m.sampling_offsets.weight.normal_(0, 0.005)
            m.attention_weights.weight.normal_(0, 0.05)
    n = 2
    shapes, starts, src, pos, ref, mask = bench._encoder_inputs(dev, n, [(800, 1333), (704, 1066)])

    def step():
        for m in layers:
            m.zero_grad()
        x = src.detach().clone().requires_grad_(True)
        y = x
        kernels = []
        for m in layers:
            y = y + m(y + pos, ref, y, shapes, starts, mask)
            kernels.append(_last())
        loss = (y * y).mean()
        loss.backward()
        torch.cuda.synchronize()
        return loss.item(), [p.grad.clone() for m in layers for p in m.parameters()], kernels

    sda._lib.set_forward_policy("adaptive")
    for _ in range(4):
        loss_a, grads_a, kern = step()
    assert kern == ["msda_rw_d32", "msda_rw_d32"], kern
    for m in layers:
        st = sda._lib.forward_policy_state(m.policy_slot)
        assert st["mode"] == 1 and 0.0 <= st["far_fraction"] < 0.45, st
    # (the backward runs on the autograd thread and the kernel names are per thread: ask the op directly, with a layer's slot)
    import MultiScaleDeformableAttention as MSDA
    m0 = layers[0]
    with torch.no_grad():
        v = m0.value_proj(src).view(n, -1, 8, 32)
        off = m0.sampling_offsets(src + pos).view(n, -1, 8, 4, 4, 2)
        lg = m0.attention_weights(src + pos).view(n, -1, 8, 16)
    MSDA.ms_deform_attn_fused_backward(v, shapes, starts, ref, off, lg, torch.ones(n, v.shape[1], 256, device=dev), mask, m0.policy_slot)
    assert _last().startswith("msda_gw_d32+"), _last()
    MSDA.ms_deform_attn_fused_backward(v, shapes, starts, ref, off, lg, torch.ones(n, v.shape[1], 256, device=dev), mask, 200)      # a slot nobody used
    assert _last().startswith("msda_bwd_gather_d32+"), _last()
    sda._lib.set_forward_policy("patch")
    loss_p, grads_p, kern_p = step()
    assert kern_p == ["msda_fwd_d32<1, 4, 408"] * 2, kern_p
    assert abs(loss_a - loss_p) <= 1e-5 * max(1.0, abs(loss_p))
    for ga, gp in zip(grads_a, grads_p):
        torch.testing.assert_close(ga, gp, rtol=2e-3, atol=2e-5 * max(1.0, float(gp.abs().max())))


def test_deterministic_algorithms_pin_the_forward_kernel():
    """torch.use_deterministic_algorithms(True): the front end asks for a forward kernel that does not depend on earlier launches
    (SEMIDETR_MSDA_FIXED_FORWARD) -- the patch kernel even under policy "window", and two passes over the same inputs
    agree bit for bit, as the reference's single kernel does."""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    value, shp, loc, attn = _case([(40, 54), (20, 27), (10, 14), (5, 7)], 2, "near", 77)
    tsh = _t(shp)
    a = (_t(value), tsh, _starts(tsh), _t(loc), _t(attn), 64)
    sda._lib.set_forward_policy("window")
    o_win = MSDA.ms_deform_attn_forward(*a)
    assert _last() == "msda_rw_d32"
    torch.use_deterministic_algorithms(True)
    try:
        o1 = MSDA.ms_deform_attn_forward(*a)
        assert _last() == "msda_fwd_d32<1, 4, 408", _last()
        sda._lib.set_forward_policy("adaptive")
        outs = [MSDA.ms_deform_attn_forward(*a) for _ in range(6)]
        assert _last() == "msda_fwd_d32<1, 4, 408", _last()
    finally:
        torch.use_deterministic_algorithms(False)
    for o in outs:
        assert torch.equal(o, o1)
    np.testing.assert_allclose(o_win.cpu().numpy(), o1.cpu().numpy(), rtol=0, atol=2e-6)
    # ... and so is the backward's gather: the lane-per-sample window gather under policy "window", the patch gather (and bitwise equal
    # small gradients from run to run) while deterministic algorithms are asked for
    gout = _t(np.random.default_rng(5).random((2, value.shape[1], M * 32)).astype(np.float32))
    sda._lib.set_forward_policy("window")
    MSDA.ms_deform_attn_backward(*a[:5], gout, 64)
    assert _last().startswith("msda_gw_d32+"), _last()
    torch.use_deterministic_algorithms(True)
    try:
        g1 = MSDA.ms_deform_attn_backward(*a[:5], gout, 64)
        assert _last().startswith("msda_bwd_gather_d32+"), _last()
        g2 = MSDA.ms_deform_attn_backward(*a[:5], gout, 64)
    finally:
        torch.use_deterministic_algorithms(False)
    assert torch.equal(g1[1], g2[1]) and torch.equal(g1[2], g2[2])


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


def test_backward_gather_follows_the_choice_made_at_forward_time():
    """under the adaptive policy the backward's small-gradient kernel (patch gather / lane-per-sample window gather) used
    to follow the slot's state at the time of the BACKWARD.  The autograd functions now record `gather_choice(slot)` right after the
    forward launch and hand it to the backward (SEMIDETR_MSDA_GATHER_WINDOW / _PATCH in `flags`): a call site whose counts change
    between its forward and its backward still runs the gather its forward knew about -- here across a mode change in both
    directions -- and the gradients equal the oracle's either way."""
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd as sda
    sda._lib.set_forward_policy("adaptive")
    shapes = [(40, 54), (20, 27), (10, 14), (5, 7)]
    shp = np.asarray(shapes, np.int64)
    tsh = _t(shp)
    tls = _starts(tsh)
    slot = 21

    def case(sigma, seed):
        value, _, ref, off, logits, gout = _encoder_case(2, shapes, sigma, seed)
        loc, attn = _prologue_np(ref, off, logits, shp, P)
        return value, loc, attn, gout

    near, far = case(1.0, 71), case(14.0, 72)

    def settle(c):
        for _ in range(8):
            MSDA.ms_deform_attn_forward(_t(c[0]), tsh, tls, _t(c[1]), _t(c[2]), 64, slot)
            torch.cuda.synchronize()

    WINDOW, PATCH = 1 << 16, 1 << 17
    # (autograd runs the backward on its own thread and semidetr_msda_last_kernels is per thread: record it where the call is made)
    ran, orig_bwd = [], MSDA.ms_deform_attn_backward

    def spy(*a):
        r = orig_bwd(*a)
        ran.append(_last())
        return r
    MSDA.ms_deform_attn_backward = spy
    try:
        _gather_choice_body(MSDA, sda, settle, near, far, tsh, tls, shp, slot, ran, WINDOW, PATCH)
    finally:
        MSDA.ms_deform_attn_backward = orig_bwd
    # a backward call that states nothing keeps the old behaviour: the slot's state now
    MSDA.ms_deform_attn_backward(_t(far[0]), tsh, tls, _t(far[1]), _t(far[2]), _t(far[3]), 64, slot)
    assert _last().startswith("msda_gw_d32+"), _last()
    MSDA.ms_deform_attn_backward(_t(far[0]), tsh, tls, _t(far[1]), _t(far[2]), _t(far[3]), 64, slot | PATCH)
    assert _last().startswith("msda_bwd_gather_d32+"), _last()
    with pytest.raises(RuntimeError, match="policy_slot"):
        MSDA.ms_deform_attn_backward(_t(far[0]), tsh, tls, _t(far[1]), _t(far[2]), _t(far[3]), 64, slot | PATCH | WINDOW)


def _gather_choice_body(MSDA, sda, settle, near, far, tsh, tls, shp, slot, ran, WINDOW, PATCH):      # noqa: N803
    settle(near)
    assert MSDA.gather_choice(slot) == WINDOW
    # the autograd function: forward while the slot says "near" ...
    tv, tl, ta = [_t(a).requires_grad_(True) for a in near[:3]]
    out = sda.MSDeformAttnFunction.apply(tv, tsh, tls, tl, ta, 64, slot)
    settle(far)                                        # ... the slot's counts move on before the backward runs
    assert MSDA.gather_choice(slot) == PATCH
    out.backward(_t(near[3]))
    assert ran[-1].startswith("msda_gw_d32+"), ran
    o_gv, o_gl, o_ga = oracle.msda_backward(near[0], shp, near[1], near[2], near[3])
    np.testing.assert_allclose(ta.grad.cpu().numpy(), o_ga, rtol=0, atol=2e-5)
    np.testing.assert_allclose(tl.grad.cpu().numpy(), o_gl, rtol=0, atol=1e-4 * max(1.0, float(np.abs(o_gl).max())))
    np.testing.assert_allclose(tv.grad.cpu().numpy(), o_gv, rtol=0, atol=1e-4 * max(1.0, float(np.abs(o_gv).max())))
    # ... and the other way round: forward while "far", the slot back to "near" before the backward
    tv, tl, ta = [_t(a).requires_grad_(True) for a in far[:3]]
    out = sda.MSDeformAttnFunction.apply(tv, tsh, tls, tl, ta, 64, slot)
    settle(near)
    assert MSDA.gather_choice(slot) == WINDOW
    out.backward(_t(far[3]))
    assert ran[-1].startswith("msda_bwd_gather_d32+"), ran
    o_gv, o_gl, o_ga = oracle.msda_backward(far[0], shp, far[1], far[2], far[3])
    np.testing.assert_allclose(ta.grad.cpu().numpy(), o_ga, rtol=0, atol=2e-5)
    np.testing.assert_allclose(tl.grad.cpu().numpy(), o_gl, rtol=0, atol=1e-4 * max(1.0, float(np.abs(o_gl).max())))
