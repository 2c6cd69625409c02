"""GPU parity: the gfx950 MSDA kernels (through MultiScaleDeformableAttention / MSDeformAttnFunction, i.e.
through the C ABI) against (a) the reference-generated fixtures, (b) the CPU oracle on seeded inputs,
(c) size-independent properties at the full BASELINE.json shapes, (d) torch.autograd.gradcheck exactly as
the reference's ops/test.py does."""
import numpy as np
import pytest
import torch

import oracle
from conftest import Golden, force_variant, kink_mask

pytestmark = pytest.mark.gpu
CASES = Golden("msda.npz").names()

# fp32 bar from BASELINE.json north_star: 1e-4 against the reference op (we hold ~1e-6)
F32_OUT_ATOL, F32_GRAD_ATOL = 2e-6, 2e-5


def _dev(*arrays):
    return [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in arrays]


def _level_start(shapes):
    hw = shapes[:, 0] * shapes[:, 1]
    return torch.cat([hw.new_zeros(1), hw.cumsum(0)[:-1]])


def _run(value, shapes, loc, attn, gout):
    import MultiScaleDeformableAttention as MSDA
    import semi_detr_amd  # noqa: F401  (installs the module above)
    v, s, lo, a, go = _dev(value, shapes, loc, attn, gout)
    ls = _level_start(s)
    out = MSDA.ms_deform_attn_forward(v, s, ls, lo, a, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, s, ls, lo, a, go, 64)
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (out, gv, gl, ga)]


@pytest.fixture(autouse=True)
def _auto_variant():
    force_variant(0, 0)
    yield
    force_variant(0, 0)


@pytest.mark.parametrize("case", CASES)
def test_hip_matches_reference_fixture(case, golden_msda):
    g = golden_msda[case]
    out, gv, gl, ga = _run(g["value"], g["shapes"], g["loc"], g["attn"], g["gout"])
    f64 = g["value"].dtype == np.float64
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=1e-14 if f64 else F32_OUT_ATOL)
    np.testing.assert_allclose(gv, g["gvalue"], rtol=0, atol=1e-13 if f64 else F32_GRAD_ATOL)
    np.testing.assert_allclose(ga, g["gattn"], rtol=0, atol=1e-13 if f64 else F32_GRAD_ATOL)
    keep = ~kink_mask(g["loc"], g["shapes"])
    np.testing.assert_allclose(gl[keep], g["gloc"][keep], rtol=0, atol=1e-12 if f64 else F32_GRAD_ATOL)


@pytest.mark.parametrize("case", CASES)
def test_hip_matches_oracle_everywhere(case, golden_msda):
    """Against the C oracle there is no excluded set: cell selection is bit-identical by construction."""
    g = golden_msda[case]
    out, gv, gl, ga = _run(g["value"], g["shapes"], g["loc"], g["attn"], g["gout"])
    o_out = oracle.msda_forward(g["value"], g["shapes"], g["loc"], g["attn"])
    o_gv, o_gl, o_ga = oracle.msda_backward(g["value"], g["shapes"], g["loc"], g["attn"], g["gout"])
    f64 = g["value"].dtype == np.float64
    np.testing.assert_allclose(out, o_out, rtol=0, atol=1e-14 if f64 else F32_OUT_ATOL)
    np.testing.assert_allclose(gv, o_gv, rtol=0, atol=1e-13 if f64 else F32_GRAD_ATOL)
    np.testing.assert_allclose(gl, o_gl, rtol=0, atol=1e-12 if f64 else F32_GRAD_ATOL)
    np.testing.assert_allclose(ga, o_ga, rtol=0, atol=1e-13 if f64 else F32_GRAD_ATOL)


def _random_case(seed, shapes, N, M, D, Lq, P, dtype, wide=True):
    rng = np.random.default_rng(seed)
    shapes = np.asarray(shapes, np.int64)
    L = len(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = (rng.random((N, S, M, D)) * 0.01).astype(dtype)
    loc = rng.random((N, Lq, M, L, P, 2))
    if wide:
        loc = loc * 1.4 - 0.2
    attn = rng.random((N, Lq, M, L, P)) + 1e-5
    attn /= attn.sum((-1, -2), keepdims=True)
    gout = rng.random((N, Lq, M * D))
    return value, shapes, loc.astype(dtype), attn.astype(dtype), gout.astype(dtype)


@pytest.mark.parametrize("variants", [(1, 8), (2, 32), (4, 8), (99, 99), (1, 808), (2, 832), (0, 0), (0, 900), (0, 901), (0, 902), (0, 910)])
@pytest.mark.parametrize("Lq", [1, 7, 8, 9, 31, 32, 33, 300])
def test_fast_path_variants_vs_oracle(variants, Lq):
    """Every forced kernel variant of the fp32 / D=32 path (forward split 1/2/4, backward 8/32 rows per
    workgroup, and the generic kernel = 99), ragged query counts around the tile sizes."""
    force_variant(*variants)
    case = _random_case(40 + Lq, [(20, 27), (10, 14), (5, 7), (3, 4)], 2, 8, 32, Lq, 4, np.float32)
    out, gv, gl, ga = _run(*case)
    o_out = oracle.msda_forward(*case[:4])
    o_gv, o_gl, o_ga = oracle.msda_backward(*case)
    np.testing.assert_allclose(out, o_out, rtol=0, atol=F32_OUT_ATOL)
    np.testing.assert_allclose(gv, o_gv, rtol=0, atol=F32_GRAD_ATOL)
    np.testing.assert_allclose(gl, o_gl, rtol=0, atol=F32_GRAD_ATOL)
    np.testing.assert_allclose(ga, o_ga, rtol=0, atol=F32_GRAD_ATOL)


@pytest.mark.parametrize("M,L,P", [(8, 5, 4), (4, 1, 1), (3, 2, 7), (16, 4, 4), (8, 4, 8)])
def test_fast_path_generic_heads_levels_points(M, L, P):
    shapes = [(17, 23), (9, 12), (5, 6), (3, 3), (2, 2)][:L]
    case = _random_case(7 * M + L + P, shapes, 2, M, 32, 45, P, np.float32)
    out, gv, gl, ga = _run(*case)
    o_out = oracle.msda_forward(*case[:4])
    o_gv, o_gl, o_ga = oracle.msda_backward(*case)
    np.testing.assert_allclose(out, o_out, rtol=0, atol=F32_OUT_ATOL)
    np.testing.assert_allclose(gv, o_gv, rtol=0, atol=F32_GRAD_ATOL)
    np.testing.assert_allclose(gl, o_gl, rtol=0, atol=F32_GRAD_ATOL)
    np.testing.assert_allclose(ga, o_ga, rtol=0, atol=F32_GRAD_ATOL)


@pytest.mark.parametrize("shapes,M,P,mode", [
    ([(20, 27), (10, 14), (5, 7), (3, 4)], 8, 4, "near"),       # DINO-like pyramid, clustered samples
    ([(20, 27), (10, 14), (5, 7), (3, 4)], 8, 4, "far"),        # same, samples anywhere (window misses)
    ([(1, 200), (3, 1)], 2, 3, "near"),                         # degenerate levels: many patches per bound
    ([(17, 9)], 3, 1, "near"),                                  # single level, ragged patches, odd head count
    ([(9, 33), (5, 17), (3, 9), (2, 5), (1, 3)], 8, 4, "wide"), # five levels, samples partly outside
    ([(20, 27), (20, 27), (19, 26)], 4, 4, "near"),             # levels of (almost) equal size: a 16 x 16 region of the
                                                                # finest level holds ~760 queries -> several passes
    ([(2, 1500), (1, 750)], 2, 4, "near"),                      # a level wider than 1423 pixels: the region scatter's 15-bit pixel
                                                                # offsets do not reach across its window -> one-by-one path (round 5)
])
@pytest.mark.parametrize("variant", [(0, 0), (1, 32), (2, 832), (408, 64), (216, 65), (804, 66), (0, 67), (500, 70), (500, 71), (0, 68), (0, 69), (0, 690), (0, 697), (0, 698), (600, 0),
                                     (700, 7000), (701, 7001), (702, 7002), (703, 7003), (704, 7004), (705, 7005), (706, 7006), (720, 0), (723, 0), (0, 920), (0, 921), (0, 922), (0, 6900), (0, 6909), (0, 6983), (0, 6984), (734, 0), (742, 0), (748, 0), (741, 0)],
                         ids=lambda v: f"f{v[0]}b{v[1]}")
def test_encoder_self_attention_vs_oracle(shapes, M, P, mode, variant):
    """num_query == spatial_size selects the patch-tiled forward and (with num_point == 4) the gather +
    owner-computes scatter backward (variant 0); (1, 32) forces the plain kernels on the same inputs; the
    other entries force the alternative patch shapes / the windowed backward.  All must match the oracle."""
    if variant[1] in (64, 65, 66, 67, 68, 69, 690, 697, 698, 6900, 6909, 6983, 6984) and (P != 4 or (variant[1] in (68, 697) and len(shapes) * P != 16)):
        pytest.skip("the windowed backward is specialised for num_point == 4")
    if variant[0] >= 700 and (P != 4 or len(shapes) not in (4, 5)):
        pytest.skip("the region-window kernels are built for num_point == 4 and 4 or 5 levels")
    force_variant(*variant)
    rng = np.random.default_rng(len(shapes) * 100 + M + P)
    shp = np.asarray(shapes, np.int64)
    L, N, D = len(shapes), 2, 32
    S = int((shp[:, 0] * shp[:, 1]).sum())
    ref = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1)
                          .reshape(-1, 2) for h, w in shapes])
    if mode == "near":
        loc = ref[None, :, None, None, None, :] + rng.standard_normal((N, S, M, L, P, 2)) * (2.0 / shp[None, None, None, :, None, ::-1])
    elif mode == "far":
        loc = rng.random((N, S, M, L, P, 2))
    else:
        loc = ref[None, :, None, None, None, :] + rng.standard_normal((N, S, M, L, P, 2)) * 0.3
    value = (rng.random((N, S, M, D)) * 0.01).astype(np.float32)
    attn = rng.random((N, S, M, L, P)) + 1e-5
    attn /= attn.sum((-1, -2), keepdims=True)
    gout = rng.random((N, S, M * D)).astype(np.float32)
    case = (value, shp, loc.astype(np.float32), attn.astype(np.float32), gout)
    out, gv, gl, ga = _run(*case)
    np.testing.assert_allclose(out, oracle.msda_forward(*case[:4]), rtol=0, atol=F32_OUT_ATOL)
    o_gv, o_gl, o_ga = oracle.msda_backward(*case)
    np.testing.assert_allclose(gv, o_gv, rtol=1e-5, atol=F32_GRAD_ATOL)
    np.testing.assert_allclose(gl, o_gl, rtol=1e-5, atol=F32_GRAD_ATOL)
    np.testing.assert_allclose(ga, o_ga, rtol=1e-5, atol=F32_GRAD_ATOL)


@pytest.mark.parametrize("D", [30, 32, 64, 71, 1025, 2048, 3096])
def test_channel_counts_of_reference_test_py_fp64(D):
    """ops/test.py:85-86 runs gradcheck for exactly these channel counts (one per CUDA dispatch branch)."""
    case = _random_case(D, [(6, 4), (3, 2)], 1, 2, D, 2, 2, np.float64, wide=False)
    out, gv, gl, ga = _run(*case)
    o_out = oracle.msda_forward(*case[:4])
    o_gv, o_gl, o_ga = oracle.msda_backward(*case)
    np.testing.assert_allclose(out, o_out, rtol=0, atol=1e-14)
    np.testing.assert_allclose(gv, o_gv, rtol=0, atol=1e-13)
    np.testing.assert_allclose(gl, o_gl, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ga, o_ga, rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("D", [30, 32, 64, 71])
def test_gradcheck_like_reference_test_py(D):
    """check_gradient_numerical of ops/test.py:63-78 (fp64 gradcheck through MSDeformAttnFunction)."""
    from semi_detr_amd import MSDeformAttnFunction
    torch.manual_seed(3)
    N, M, Lq, L, P = 1, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long).cuda()
    ls = _level_start(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = (torch.rand(N, S, M, D).cuda() * 0.01).double().requires_grad_(True)
    loc = torch.rand(N, Lq, M, L, P, 2).cuda().double().requires_grad_(True)
    attn = torch.rand(N, Lq, M, L, P).cuda() + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
    assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, ls, loc, attn, 2))


@pytest.mark.parametrize("D", [1025, 2048, 3096])
def test_gradcheck_wide_channels_like_reference_test_py(D):
    """ops/test.py:85-86 also runs check_gradient_numerical for 1025, 2048 and 3096 channels (the CUDA op's multi-block
    reduction branches).  Same fp64 gradcheck through MSDeformAttnFunction on a pyramid small enough that the numerical
    Jacobian (2 evaluations per input element) stays cheap."""
    from semi_detr_amd import MSDeformAttnFunction
    torch.manual_seed(D)
    N, M, Lq, L, P = 1, 1, 2, 2, 2
    shapes = torch.as_tensor([(2, 2), (1, 1)], dtype=torch.long).cuda()
    ls = _level_start(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = (torch.rand(N, S, M, D).cuda() * 0.01).double().requires_grad_(True)
    loc = torch.rand(N, Lq, M, L, P, 2).cuda().double().requires_grad_(True)
    attn = torch.rand(N, Lq, M, L, P).cuda() + 1e-5
    attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).double().requires_grad_(True)
    assert torch.autograd.gradcheck(MSDeformAttnFunction.apply, (value, shapes, ls, loc, attn, 2))


def test_reference_test_py_forward_checks():
    """check_forward_equal_with_pytorch_{double,float} of ops/test.py:31-60, with the oracle in the role of
    ms_deform_attn_core_pytorch (itself pinned to it by the fixtures)."""
    from semi_detr_amd import MSDeformAttnFunction
    torch.manual_seed(3)
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long).cuda()
    ls = _level_start(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    for dt, kw in ((torch.float64, {}), (torch.float32, dict(rtol=1e-2, atol=1e-3))):
        value = (torch.rand(N, S, M, D).cuda() * 0.01).to(dt)
        loc = torch.rand(N, Lq, M, L, P, 2).cuda().to(dt)
        attn = torch.rand(N, Lq, M, L, P).cuda() + 1e-5
        attn = (attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)).to(dt)
        out = MSDeformAttnFunction.apply(value, shapes, ls, loc, attn, 2).cpu()
        want = torch.from_numpy(oracle.msda_forward(value.cpu().numpy(), shapes.cpu().numpy(),
                                                    loc.cpu().numpy(), attn.cpu().numpy()))
        assert torch.allclose(out, want, **kw)
        if dt == torch.float32:   # and our own, much tighter, bar
            assert (out - want).abs().max() < 1e-6


# ---- full BASELINE.json shapes --------------------------------------------------------------------
LEVELS = [(100, 167), (50, 84), (25, 42), (13, 21)]      # 800x1333 input, S = 22223


def test_microbench_shape_vs_oracle():
    """N=2, Lq=300, M=8, D=32, L=4, P=4, S=22223 (BASELINE.json metric shape): compare with the oracle."""
    case = _random_case(3, LEVELS, 2, 8, 32, 300, 4, np.float32, wide=False)
    out, gv, gl, ga = _run(*case)
    o_out = oracle.msda_forward(*case[:4])
    o_gv, o_gl, o_ga = oracle.msda_backward(*case)
    assert np.abs(out - o_out).max() < 1e-6            # north-star bar is 1e-4
    assert np.abs(gv - o_gv).max() < 1e-5
    assert np.abs(gl - o_gl).max() < 1e-4 * max(1.0, np.abs(o_gl).max())
    assert np.abs(ga - o_ga).max() < 1e-5


def test_microbench_shape_vs_reference_fixture():
    """The same shape against the REFERENCE's own CPU path (tests/golden/msda_full.npz; inputs regenerated from
    the seed exactly as oracle/gen_golden.py:full_inputs does)."""
    import os
    from conftest import FULL_LEVELS, check_full_shape, full_shape_inputs
    golden = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msda_full.npz"))
    value, loc, attn, gout = [t.numpy() for t in full_shape_inputs()]
    out, gv, gl, ga = _run(value, np.asarray(FULL_LEVELS, np.int64), loc, attn, gout)
    check_full_shape(out, gv, gl, ga, loc, golden)


@pytest.mark.parametrize("Lq", [300, 1100, 22223])
def test_full_size_properties(Lq):
    """Size-independent properties at decoder and encoder scale (bs=2):
    constant value map + in-bounds interior samples -> output == constant (weights sum to 1);
    linearity in value; sum(grad_value) == sum_k a_k * g . (sum of valid corner weights) (mass conservation);
    grad_attn == <g, sampled value>."""
    import MultiScaleDeformableAttention as MSDA
    torch.manual_seed(Lq)
    N, M, D, L, P = 2, 8, 32, 4, 4
    shapes = torch.as_tensor(LEVELS, dtype=torch.long).cuda()
    ls = _level_start(shapes)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    # interior samples: keep one pixel away from the border so all 4 corners are valid (robust to rounding)
    lo_xy = torch.tensor([[1.0 / w, 1.0 / h] for h, w in LEVELS]).cuda().view(1, 1, 1, L, 1, 2)
    loc = lo_xy + torch.rand(N, Lq, M, L, P, 2).cuda() * (1 - 2 * lo_xy) * 0.999
    attn = torch.rand(N, Lq, M, L, P).cuda() + 1e-5
    attn = attn / attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
    const = torch.full((N, S, M, D), 0.37).cuda()
    out = MSDA.ms_deform_attn_forward(const, shapes, ls, loc, attn, 64)
    assert (out - 0.37).abs().max() < 1e-5
    value = torch.rand(N, S, M, D).cuda()
    o1 = MSDA.ms_deform_attn_forward(value, shapes, ls, loc, attn, 64)
    o2 = MSDA.ms_deform_attn_forward(value * 2, shapes, ls, loc, attn, 64)
    assert torch.allclose(o2, 2 * o1, rtol=1e-6, atol=1e-6)
    g = torch.rand(N, Lq, M * D).cuda()
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, shapes, ls, loc, attn, g, 64)
    # mass conservation per (n, m, channel): all corner weights sum to 1 for interior samples
    want = (g.view(N, Lq, M, D).double() * attn.sum((-1, -2)).double()[..., None]).sum(1)     # (N, M, D)
    got = gv.double().sum(1)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-3)
    # <grad_attn, attn> == <g, out>   (out is linear in attn)
    lhs = (ga.double() * attn.double()).sum()
    rhs = (g.double() * o1.double()).sum()
    assert abs(lhs - rhs) / abs(rhs) < 1e-5
    # the constant map has zero spatial gradient
    _, gl_c, _ = MSDA.ms_deform_attn_backward(const, shapes, ls, loc, attn, g, 64)
    assert gl_c.abs().max() < 1e-4
    assert torch.isfinite(gl).all()


def test_encoder_shape_vs_oracle_one_image():
    """Encoder scale (Lq = S = 22223, one image) against the oracle; locations = pixel centre + N(0, 2px)."""
    rng = np.random.default_rng(5)
    N, M, D, L, P = 1, 8, 32, 4, 4
    shapes = np.asarray(LEVELS, np.int64)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    ref = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1)
                          .reshape(-1, 2) for h, w in LEVELS])                    # (S, 2) x,y
    loc = ref[None, :, None, None, None, :] + rng.standard_normal((N, S, M, L, P, 2)) * \
        (2.0 / shapes[None, None, None, :, None, ::-1])
    value = (rng.random((N, S, M, D)) * 0.01).astype(np.float32)
    attn = rng.random((N, S, M, L, P)) + 1e-5
    attn /= attn.sum((-1, -2), keepdims=True)
    gout = rng.random((N, S, M * D)).astype(np.float32)
    case = (value, shapes, loc.astype(np.float32), attn.astype(np.float32), gout)
    out, gv, gl, ga = _run(*case)
    o_out = oracle.msda_forward(*case[:4])
    o_gv, o_gl, o_ga = oracle.msda_backward(*case)
    assert np.abs(out - o_out).max() < 1e-6
    assert np.abs(gv - o_gv).max() < 1e-4 * max(1.0, np.abs(o_gv).max())
    assert np.abs(gl - o_gl).max() < 1e-4 * max(1.0, np.abs(o_gl).max())
    assert np.abs(ga - o_ga).max() < 1e-5


def test_precondition_errors_match_reference():
    """ms_deform_attn_cuda.cu:28-52 asserts and ms_deform_attn.h:38 CPU error -> RuntimeError."""
    import MultiScaleDeformableAttention as MSDA
    case = _random_case(1, [(6, 4), (3, 2)], 3, 2, 32, 5, 2, np.float32)
    v, s, lo, a, go = _dev(*case)
    ls = _level_start(s)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(v.cpu(), s.cpu(), ls.cpu(), lo.cpu(), a.cpu(), 64)
    with pytest.raises(RuntimeError, match="value tensor has to be contiguous"):
        MSDA.ms_deform_attn_forward(v.transpose(2, 3), s, ls, lo, a, 64)
    with pytest.raises(RuntimeError, match="must divide im2col_step"):
        MSDA.ms_deform_attn_forward(v, s, ls, lo, a, 2)          # batch 3 % min(3, 2) != 0
    with pytest.raises(RuntimeError, match="not implemented for 'Half'"):
        MSDA.ms_deform_attn_forward(v.half(), s, ls, lo.half(), a.half(), 64)
    with pytest.raises(RuntimeError, match="spatial_shapes must be a CUDA tensor"):
        MSDA.ms_deform_attn_forward(v, s.cpu(), ls, lo, a, 64)
    out = MSDA.ms_deform_attn_forward(v, s, ls, lo, a, 3)
    assert out.shape == (3, 5, 64)


def test_num_query_equal_spatial_size_without_pixel_queries():
    """`num_query == spatial_size` alone must not select the self-attention kernels.  Here the value map has
    five rows more than the levels cover (never sampled) and Lq == S by coincidence: the level table does not tile
    [0, S), so the front end must not set SEMIDETR_MSDA_QUERIES_ARE_PIXELS, every output row must be written, and the
    result must match the oracle (the reference op has no coupling between queries and pixels)."""
    import MultiScaleDeformableAttention as MSDA
    rng = np.random.default_rng(77)
    shapes = np.asarray([(6, 9), (3, 5)], np.int64)
    N, M, D, L, P = 2, 8, 32, 2, 4
    S = int((shapes[:, 0] * shapes[:, 1]).sum()) + 5
    Lq = S
    value = rng.random((N, S, M, D)).astype(np.float32)
    loc = (rng.random((N, Lq, M, L, P, 2)) * 1.2 - 0.1).astype(np.float32)
    attn = rng.random((N, Lq, M, L, P)).astype(np.float32)
    gout = rng.random((N, Lq, M * D)).astype(np.float32)
    v, s, lo, a, go = _dev(value, shapes, loc, attn, gout)
    ls = _level_start(s)
    assert MSDA.pyramid_check(s, ls, S) == 0
    out = MSDA.ms_deform_attn_forward(v, s, ls, lo, a, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, s, ls, lo, a, go, 64)
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.cpu().numpy(), oracle.msda_forward(value, shapes, loc, attn), rtol=0, atol=F32_OUT_ATOL)
    o_gv, o_gl, o_ga = oracle.msda_backward(value, shapes, loc, attn, gout)
    np.testing.assert_allclose(gv.cpu().numpy(), o_gv, rtol=0, atol=F32_GRAD_ATOL)
    np.testing.assert_allclose(gl.cpu().numpy(), o_gl, rtol=1e-5, atol=F32_GRAD_ATOL)
    np.testing.assert_allclose(ga.cpu().numpy(), o_ga, rtol=1e-5, atol=F32_GRAD_ATOL)
    assert float(gv[:, -5:].abs().max()) == 0.0        # rows no level covers receive no gradient


@pytest.mark.parametrize("L,P,Lq,N,Mh", [(8, 8, 300, 2, 2), (8, 8, 2100, 2, 8), (16, 8, 2100, 2, 8), (16, 16, 2100, 2, 8),
                                          (9, 4, 700, 1, 8), (15, 4, 2100, 2, 8), (16, 4, 0, 1, 4), (6, 4, 0, 2, 8)])
def test_wide_level_point_products_vs_oracle(L, P, Lq, N, Mh):
    """num_levels * num_point up to 256 stays on the D = 32 path: the records of a workgroup's rows have to fit its LDS, so the
    dispatcher cuts the rows per workgroup (forward split, 8-row strips backward) and keeps the merged launch for L * P <= 36 --
    round 3 found `merged launch too large` / `invalid argument` errors here (8 levels x 8 points at 600 queries; 16 x 16 at
    large launches)."""
    import MultiScaleDeformableAttention as MSDA
    rng = np.random.default_rng(L * P + Lq)
    shapes = [(max(2, 12 - l), max(2, 14 - l)) for l in range(L)]
    shp = np.asarray(shapes, np.int64)
    S = int((shp[:, 0] * shp[:, 1]).sum())
    Lq = Lq or S                                   # 0: encoder self-attention (queries are the pixels), many levels
    value = rng.random((N, S, Mh, 32)).astype(np.float32)
    loc = (rng.random((N, Lq, Mh, L, P, 2)) * 1.2 - 0.1).astype(np.float32)
    attn = rng.random((N, Lq, Mh, L, P)).astype(np.float32)
    attn /= attn.sum((-1, -2), keepdims=True)
    gout = rng.standard_normal((N, Lq, Mh * 32)).astype(np.float32)
    v, s, lo, a, go = _dev(value, shp, loc, attn, gout)
    ls = _level_start(s)
    out = MSDA.ms_deform_attn_forward(v, s, ls, lo, a, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(v, s, ls, lo, a, go, 64)
    torch.cuda.synchronize()
    o_gv, o_gl, o_ga = oracle.msda_backward(value, shp, loc, attn, gout)
    np.testing.assert_allclose(out.cpu().numpy(), oracle.msda_forward(value, shp, loc, attn), rtol=0, atol=F32_OUT_ATOL)
    np.testing.assert_allclose(gv.cpu().numpy(), o_gv, rtol=0, atol=F32_GRAD_ATOL * max(1.0, float(np.abs(o_gv).max())))
    np.testing.assert_allclose(gl.cpu().numpy(), o_gl, rtol=1e-5, atol=F32_GRAD_ATOL * max(1.0, float(np.abs(o_gl).max())))
    np.testing.assert_allclose(ga.cpu().numpy(), o_ga, rtol=1e-5, atol=F32_GRAD_ATOL * max(1.0, float(np.abs(o_ga).max())))


def _torch_msda(value, shapes, starts, loc, attn):
    """Plain PyTorch fp32 statement of the op for an ARBITRARY level table (levels may leave gaps or overlap): bilinear
    sampling by explicit corner gathers, zero padding outside a level (ms_deform_im2col_cuda.cuh:33-85, 237-299)."""
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = value.new_zeros(N, Lq, M, D)
    n_i = torch.arange(N, device=value.device).view(N, 1, 1, 1)
    m_i = torch.arange(M, device=value.device).view(1, 1, M, 1)
    for l, (H, W) in enumerate(shapes):
        x = loc[:, :, :, l, :, 0] * W - 0.5                 # (N, Lq, M, P)
        y = loc[:, :, :, l, :, 1] * H - 0.5
        x0, y0 = torch.floor(x), torch.floor(y)
        lx, ly = x - x0, y - y0
        for dy, dx, w in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
            xi, yi = (x0 + dx).long(), (y0 + dy).long()
            ok = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            row = starts[l] + yi.clamp(0, H - 1) * W + xi.clamp(0, W - 1)
            v = value[n_i, row, m_i]                       # (N, Lq, M, P, D)
            out = out + (v * (w * ok * attn[:, :, :, l, :])[..., None]).sum(3)
    return out.reshape(N, Lq, M * D)


@pytest.mark.parametrize("starts,S", [([0, 30, 500, 590], 640),       # levels 0 and 1 overlap (rows 30 .. 419), a gap before level 2
                                      ([0, 420, 420, 600], 700),      # levels 1 and 2 share their first rows
                                      ([0, 420, 540, 575], 590)])     # the canonical table (control)
@pytest.mark.parametrize("Lq", [40, 300])
def test_level_tables_with_overlapping_levels(starts, S, Lq):
    """level_start_index is an input like any other: levels that alias the same value rows are legal for the reference (its
    atomicAdd sums both levels' contributions).  The exclusive store form of the merged backward (one chunk of queries: Lq <= 384)
    must notice the aliasing and fall back to atomics."""
    import MultiScaleDeformableAttention as MSDA
    shapes = [(20, 21), (10, 12), (5, 7), (3, 5)]
    N, M, D, P, L = 2, 4, 32, 4, 4
    g = torch.Generator(device="cuda").manual_seed(S + Lq)
    value = torch.rand(N, S, M, D, generator=g, device="cuda")
    loc = torch.rand(N, Lq, M, L, P, 2, generator=g, device="cuda") * 1.2 - 0.1
    attn = torch.rand(N, Lq, M, L, P, generator=g, device="cuda") + 1e-3
    attn = attn / attn.sum((-1, -2), keepdim=True)
    gout = torch.randn(N, Lq, M * D, generator=g, device="cuda")
    tsh = torch.tensor(shapes, device="cuda")
    tls = torch.tensor(starts, device="cuda")
    out = MSDA.ms_deform_attn_forward(value, tsh, tls, loc, attn, 64)
    gv, gl, ga = MSDA.ms_deform_attn_backward(value, tsh, tls, loc, attn, gout, 64)
    v2, a2 = value.clone().requires_grad_(True), attn.clone().requires_grad_(True)
    ref = _torch_msda(v2, shapes, starts, loc, a2)
    ref.backward(gout)
    torch.testing.assert_close(out, ref.detach(), rtol=0, atol=5e-6)
    torch.testing.assert_close(gv, v2.grad, rtol=0, atol=5e-5)
    torch.testing.assert_close(ga, a2.grad, rtol=1e-5, atol=5e-5)
