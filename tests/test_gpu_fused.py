"""GPU parity of the fused MSDeformAttn prologue/epilogue: the kernels that
consume reference points + RAW offsets + RAW logits must reproduce, to fp32 rounding, (a) the oracle applied
to a numpy restatement of ms_deform_attn.py:99-111 and (b) the op-by-op path (torch softmax / location
arithmetic + MSDeformAttnFunction), forward and all gradients."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _prologue_np(ref, off, logits, shapes, P):
    """ms_deform_attn.py:99-111 in numpy (float64 accumulate, cast back)."""
    N, Lq, M, L, _, _ = off.shape
    lg = logits.astype(np.float64)
    lg = lg - lg.max(-1, keepdims=True)
    a = np.exp(lg)
    a = (a / a.sum(-1, keepdims=True)).reshape(N, Lq, M, L, P)
    if ref.shape[-1] == 2:
        norm = np.stack([shapes[:, 1], shapes[:, 0]], -1).astype(np.float64)
        loc = ref[:, :, None, :, None, :].astype(np.float64) + off / norm[None, None, None, :, None, :]
    else:
        loc = ref[:, :, None, :, None, :2].astype(np.float64) + off.astype(np.float64) / P * ref[:, :, None, :, None, 2:] * 0.5
    return loc.astype(np.float32), a.astype(np.float32)


def _case(seed, shapes, N, M, Lq, P, ref_dim, encoder=False):
    rng = np.random.default_rng(seed)
    shp = np.asarray(shapes, np.int64)
    L = len(shapes)
    S = int((shp[:, 0] * shp[:, 1]).sum())
    if encoder:
        Lq = S
        ref2 = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1)
                               .reshape(-1, 2) for h, w in shapes])
        ref = np.broadcast_to(ref2[None, :, None, :], (N, S, L, 2)).copy()
        if ref_dim == 4:
            ref = np.concatenate([ref, rng.random((N, S, L, 2)) * 0.2 + 0.02], -1)
        off = rng.standard_normal((N, Lq, M, L, P, 2)) * 2.5
    else:
        ref = rng.random((N, Lq, L, ref_dim))
        if ref_dim == 4:
            ref[..., 2:] = ref[..., 2:] * 0.3 + 0.02
        off = rng.standard_normal((N, Lq, M, L, P, 2)) * (3.0 if ref_dim == 2 else 1.5)
    value = (rng.random((N, S, M, 32)) * 0.01).astype(np.float32)
    logits = (rng.standard_normal((N, Lq, M, L * P)) * 2).astype(np.float32)
    gout = rng.random((N, Lq, M * 32)).astype(np.float32)
    return value, shp, ref.astype(np.float32), off.astype(np.float32), logits, gout


CASES = [
    ("dec_ref4", [(20, 27), (10, 14), (5, 7), (3, 4)], 2, 8, 70, 4, 4, False),
    ("dec_ref2", [(20, 27), (10, 14), (5, 7), (3, 4)], 2, 8, 33, 4, 2, False),
    ("enc_ref2", [(20, 27), (10, 14), (5, 7), (3, 4)], 2, 8, 0, 4, 2, True),     # Lq == S: patch fwd + window bwd
    ("enc_ref4", [(9, 33), (5, 17), (3, 9)], 1, 4, 0, 4, 4, True),
    ("odd", [(7, 5), (3, 3)], 2, 3, 11, 3, 2, False),                           # M=3, P=3: strips kernels
]


@pytest.mark.parametrize("name,shapes,N,M,Lq,P,ref_dim,enc", CASES)
def test_fused_matches_oracle_and_unfused(name, shapes, N, M, Lq, P, ref_dim, enc):
    from semi_detr_amd import MSDeformAttnFunction, MSDeformAttnFusedFunction
    value, shp, ref, off, logits, gout = _case(len(name) + N + M, shapes, N, M, Lq, P, ref_dim, enc)
    L = len(shapes)
    dev = "cuda"
    tsh = torch.from_numpy(shp).to(dev)
    tls = torch.cat([tsh.new_zeros(1), (tsh[:, 0] * tsh[:, 1]).cumsum(0)[:-1]])
    tg = torch.from_numpy(gout).to(dev)

    def leaves():
        return [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (value, ref, off, logits)]

    # fused
    v1, r1, o1, l1 = leaves()
    out1 = MSDeformAttnFusedFunction.apply(v1, tsh, tls, r1, o1, l1)
    out1.backward(tg)
    # op by op (the reference module's sequence)
    v2, r2, o2, l2 = leaves()
    w2 = torch.softmax(l2, -1).view(*l2.shape[:3], L, P)
    if ref_dim == 2:
        norm = torch.stack([tsh[:, 1], tsh[:, 0]], -1)
        loc2 = r2[:, :, None, :, None, :] + o2 / norm[None, None, None, :, None, :]
    else:
        loc2 = r2[:, :, None, :, None, :2] + o2 / P * r2[:, :, None, :, None, 2:] * 0.5
    out2 = MSDeformAttnFunction.apply(v2, tsh, tls, loc2.contiguous(), w2.contiguous(), 64)
    out2.backward(tg)
    torch.cuda.synchronize()

    # (a) against the oracle on the numpy prologue
    loc_np, a_np = _prologue_np(ref, off, logits, shp, P)
    o_out = oracle.msda_forward(value, shp, loc_np, a_np)
    np.testing.assert_allclose(out1.detach().cpu().numpy(), o_out, rtol=0, atol=5e-6)
    o_gv, _, _ = oracle.msda_backward(value, shp, loc_np, a_np, gout)
    np.testing.assert_allclose(v1.grad.cpu().numpy(), o_gv, rtol=1e-4, atol=5e-5)
    # (b) against the op-by-op path on the GPU (same kernels underneath, different prologue)
    np.testing.assert_allclose(out1.detach().cpu().numpy(), out2.detach().cpu().numpy(), rtol=0, atol=5e-6)
    scale = lambda t: max(1.0, float(t.abs().max()))
    for a, b, what in ((v1.grad, v2.grad, "value"), (o1.grad, o2.grad, "offsets"), (l1.grad, l2.grad, "logits"),
                       (r1.grad, r2.grad, "reference_points")):
        assert a is not None and b is not None, what
        err = float((a - b).abs().max())
        assert err < 2e-4 * scale(b), (what, err, scale(b))


def test_module_fused_equals_unfused_and_reference_fixture(golden_module):
    """MSDeformAttn with fuse_prologue True / False on the DINO shape, and the fused fp32 module against the
    reference module's own fp64 fixture."""
    from semi_detr_amd import MSDeformAttn
    torch.manual_seed(0)
    m = MSDeformAttn(256, 4, 8, 4).cuda()
    with torch.no_grad():
        m.sampling_offsets.weight.normal_(0, 0.02)
        m.attention_weights.weight.normal_(0, 0.1)
    levels = [(20, 27), (10, 14), (5, 7), (3, 4)]
    shapes = torch.as_tensor(levels, dtype=torch.long).cuda()
    ls = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    S = sum(h * w for h, w in levels)
    mask = torch.rand(2, S).cuda() < 0.1
    res = {}
    for fused in (True, False):
        m.fuse_prologue = fused
        m.zero_grad()
        q = torch.randn(2, 50, 256, generator=torch.Generator().manual_seed(1)).cuda().requires_grad_(True)
        src = torch.randn(2, S, 256, generator=torch.Generator().manual_seed(2)).cuda().requires_grad_(True)
        ref = torch.rand(2, 50, 4, 4, generator=torch.Generator().manual_seed(3)).cuda()
        ref[..., 2:] = ref[..., 2:] * 0.2 + 0.05
        out = m(q, ref, src, shapes, ls, mask)
        out.square().sum().backward()
        res[fused] = [out.detach(), q.grad, src.grad, m.sampling_offsets.weight.grad.clone(),
                      m.attention_weights.weight.grad.clone(), m.value_proj.weight.grad.clone()]
    for a, b in zip(res[True], res[False]):
        assert float((a - b).abs().max()) < 1e-4 * max(1.0, float(b.abs().max()))
    # the reference module's fixture (fp64) -> fused fp32 module with the same weights; 8 channels per head is
    # not the fused configuration, so this checks the op-by-op fallback of the same module object
    g = golden_module["ref4"]
    m2 = MSDeformAttn(d_model=32, n_levels=3, n_heads=4, n_points=2)
    m2.load_state_dict({k[3:]: torch.from_numpy(v).float() for k, v in g.items() if k.startswith("sd.")})
    m2 = m2.cuda()
    sh = torch.from_numpy(g["shapes"]).cuda()
    ls2 = torch.cat([sh.new_zeros(1), (sh[:, 0] * sh[:, 1]).cumsum(0)[:-1]])
    out = m2(torch.from_numpy(g["query"]).float().cuda(), torch.from_numpy(g["ref"]).float().cuda(),
             torch.from_numpy(g["src"]).float().cuda(), sh, ls2, torch.from_numpy(g["mask"]).cuda())
    np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=1e-4, atol=1e-5)


def test_fused_errors():
    import MultiScaleDeformableAttention as MSDA
    v = torch.zeros(1, 4, 2, 32).cuda()
    sh = torch.tensor([[2, 2]]).cuda()
    ls = torch.tensor([0]).cuda()
    with pytest.raises(ValueError, match="Last dim of reference_points must be 2 or 4"):
        MSDA.ms_deform_attn_fused_forward(v, sh, ls, torch.zeros(1, 3, 1, 3).cuda(),
                                          torch.zeros(1, 3, 2, 1, 1, 2).cuda(), torch.zeros(1, 3, 2, 1).cuda())
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_fused_forward(v.cpu(), sh.cpu(), ls.cpu(), torch.zeros(1, 3, 1, 2),
                                          torch.zeros(1, 3, 2, 1, 1, 2), torch.zeros(1, 3, 2, 1))
    with pytest.raises(RuntimeError, match="only channels == 32"):
        MSDA.ms_deform_attn_fused_forward(torch.zeros(1, 4, 2, 16).cuda(), sh, ls, torch.zeros(1, 3, 1, 2).cuda(),
                                          torch.zeros(1, 3, 2, 1, 1, 2).cuda(), torch.zeros(1, 3, 2, 1).cuda())


def test_fused_reference_points_view_at_an_8_byte_offset():
    """The kernels read a reference point with one 16-byte load; a contiguous batch slice of 2-d points with an odd Lq * L starts
    8 bytes off -- the front end hands the library an aligned copy, and the last point's load runs into the buffer bound."""
    import MultiScaleDeformableAttention as MSDA
    torch.manual_seed(5)
    N, Lq, M, L, P = 2, 3, 2, 1, 2
    sh = torch.tensor([[3, 4]]).cuda()
    ls = torch.tensor([0]).cuda()
    v = torch.rand(N, 12, M, 32).cuda()
    ref3 = torch.rand(N + 1, Lq, L, 2).cuda()
    ref = ref3[1:]                                     # contiguous, data_ptr 24 bytes into the allocation
    assert ref.is_contiguous() and ref.data_ptr() % 16 != 0
    off = torch.randn(N, Lq, M, L, P, 2).cuda()
    lg = torch.randn(N, Lq, M, L * P).cuda()
    out = MSDA.ms_deform_attn_fused_forward(v, sh, ls, ref, off, lg)
    want = MSDA.ms_deform_attn_fused_forward(v, sh, ls, ref.clone(), off, lg)
    assert torch.equal(out, want)
    go = torch.rand_like(out)
    got = MSDA.ms_deform_attn_fused_backward(v, sh, ls, ref, off, lg, go)
    exp = MSDA.ms_deform_attn_fused_backward(v, sh, ls, ref.clone(), off, lg, go)
    for a, b in zip(got, exp):
        assert torch.allclose(a, b, rtol=0, atol=1e-6)


MASK_CASES = [
    ("dec_ref4", [(20, 27), (10, 14), (5, 7), (3, 4)], 2, 8, 70, 4, 4, False),      # strips backward
    ("dec_big", [(20, 27), (10, 14), (5, 7), (3, 4)], 2, 8, 300, 4, 4, False),      # N * Lq >= 512: merged level scatter
    ("enc_ref2", [(20, 27), (10, 14), (5, 7), (3, 4)], 2, 8, 0, 4, 2, True),        # patch forward, gather + region scatter
    ("enc_5lvl", [(9, 33), (5, 17), (3, 9), (2, 5), (1, 3)], 2, 8, 0, 4, 2, True),  # L * P = 20 gather
    ("odd", [(7, 5), (3, 3)], 2, 3, 11, 3, 2, False),                               # M = 3, P = 3
]


@pytest.mark.parametrize("name,shapes,N,M,Lq,P,ref_dim,enc", MASK_CASES)
def test_fused_padding_mask_inside_the_kernels(name, shapes, N, M, Lq, P, ref_dim, enc):
    """`value.masked_fill(input_padding_mask[..., None], 0)` (ms_deform_attn.py:95-96) folded into the fused kernels
    (VERDICT r02 #4): the op is handed the UNMASKED value + the mask and must equal the oracle on the masked value -- the
    forward, the two small gradients, and grad_value, whose rows at padded pixels must be EXACTLY zero (that is the backward
    of masked_fill).  Padding = the right / bottom band of every level, like an image smaller than the batch canvas."""
    from semi_detr_amd import MSDeformAttnFusedFunction
    value, shp, ref, off, logits, gout = _case(3 * len(name) + N, shapes, N, M, Lq, P, ref_dim, enc)
    L = len(shapes)
    rng = np.random.default_rng(len(name))
    masks = []
    for h, w in shapes:
        mk = np.zeros((N, h, w), bool)
        for n in range(N):
            vh, vw = rng.integers(max(1, h // 2), h + 1), rng.integers(max(1, w // 2), w + 1)
            mk[n, vh:, :] = True
            mk[n, :, vw:] = True
        masks.append(mk.reshape(N, -1))
    mask = np.concatenate(masks, 1)
    assert 0 < mask.mean() < 0.8
    value[:, :, 0, 0][mask] = np.nan          # a padded pixel may hold anything: it must never be read into a result
    loc, attn = _prologue_np(ref, off, logits, shp, P)
    vm = np.where(mask[:, :, None, None], 0.0, value).astype(np.float32)
    o_out = oracle.msda_forward(vm, shp, loc, attn)
    o_gv, o_gl, o_ga = oracle.msda_backward(vm, shp, loc, attn, gout)
    o_gv = np.where(mask[:, :, None, None], 0.0, o_gv)           # masked_fill's backward
    dev = "cuda"
    tsh = torch.from_numpy(shp).to(dev)
    tls = torch.cat([tsh.new_zeros(1), (tsh[:, 0] * tsh[:, 1]).cumsum(0)[:-1]])
    tv, tr, to, tl = [torch.from_numpy(a).to(dev).requires_grad_(True) for a in (value, ref, off, logits)]
    out = MSDeformAttnFusedFunction.apply(tv, tsh, tls, tr, to, tl, torch.from_numpy(mask).to(dev))
    out.backward(torch.from_numpy(gout).to(dev))
    np.testing.assert_allclose(out.detach().cpu().numpy(), o_out, rtol=0, atol=2e-6)
    gv = tv.grad.cpu().numpy()
    assert np.all(gv[mask] == 0.0), "padded pixels must receive exactly zero gradient"
    np.testing.assert_allclose(gv, o_gv, rtol=1e-5, atol=2e-5)
    if ref_dim == 2:
        scale = 1.0 / np.stack([shp[:, 1], shp[:, 0]], -1).astype(np.float64)[None, None, None, :, None, :]
    else:
        scale = 0.5 * ref[:, :, None, :, None, 2:].astype(np.float64) / P
    want_off = (o_gl.astype(np.float64) * scale).astype(np.float32)
    a64 = attn.astype(np.float64).reshape(N, -1, M, L * P)
    g64 = o_ga.astype(np.float64).reshape(N, -1, M, L * P)
    want_log = (a64 * (g64 - (a64 * g64).sum(-1, keepdims=True))).astype(np.float32)
    from conftest import kink_mask
    ok = ~kink_mask(loc, shp)
    np.testing.assert_allclose(to.grad.cpu().numpy()[ok], want_off[ok], rtol=1e-4, atol=2e-5 * max(1.0, float(np.abs(want_off).max())))
    np.testing.assert_allclose(tl.grad.cpu().numpy(), want_log, rtol=1e-4, atol=2e-5)
