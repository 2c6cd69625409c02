"""CPU: the C oracle (oracle/msda_oracle.c) against fixtures generated from the reference's own
ms_deform_attn_core_pytorch + autograd (oracle/gen_golden.py)."""
import numpy as np
import pytest

import oracle
from conftest import Golden, check_full_shape, full_shape_inputs, kink_mask, skipped_sample_mask, FULL_LEVELS

CASES = Golden("msda.npz").names()


@pytest.mark.parametrize("case", CASES)
def test_oracle_forward_matches_reference(case, golden_msda):
    g = golden_msda[case]
    out = oracle.msda_forward(g["value"], g["shapes"], g["loc"], g["attn"])
    tol = 1e-15 if g["value"].dtype == np.float64 else 2e-8
    assert out.dtype == g["out"].dtype and out.shape == g["out"].shape
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=tol)


@pytest.mark.parametrize("case", CASES)
def test_oracle_backward_matches_reference(case, golden_msda):
    g = golden_msda[case]
    gv, gl, ga = oracle.msda_backward(g["value"], g["shapes"], g["loc"], g["attn"], g["gout"])
    f64 = g["value"].dtype == np.float64
    np.testing.assert_allclose(gv, g["gvalue"], rtol=0, atol=1e-14 if f64 else 5e-7)
    np.testing.assert_allclose(ga, g["gattn"], rtol=0, atol=1e-14 if f64 else 5e-7)
    keep = ~kink_mask(g["loc"], g["shapes"])
    assert keep.mean() > 0.05
    np.testing.assert_allclose(gl[keep], g["gloc"][keep], rtol=0, atol=1e-13 if f64 else 5e-6)
    # exactly on the skip boundary the CUDA semantics (strict inequalities) give exactly zero
    assert np.all(gl[skipped_sample_mask(g["loc"], g["shapes"])] == 0)


def test_oracle_zero_outside_and_linearity():
    rng = np.random.default_rng(0)
    shapes = np.array([[5, 4], [2, 3]], np.int64)
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = rng.random((1, S, 2, 8)).astype(np.float64)
    loc = rng.random((1, 6, 2, 2, 3, 2)) + 2.0          # everything far outside -> all samples skipped
    attn = rng.random((1, 6, 2, 2, 3))
    assert np.all(oracle.msda_forward(value, shapes, loc, attn) == 0)
    loc = rng.random((1, 6, 2, 2, 3, 2))
    a = oracle.msda_forward(value, shapes, loc, attn)
    b = oracle.msda_forward(2 * value, shapes, loc, attn)
    np.testing.assert_allclose(b, 2 * a, rtol=1e-14)


def test_oracle_full_microbench_shape_matches_reference():
    """The BASELINE.json metric shape itself (N=2, Lq=300, S=22223): inputs regenerated from the seed, outputs of the
    reference's CPU path committed whole (out, grad_loc, grad_attn) / sampled + per-level sums (grad_value)."""
    import os
    golden = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "msda_full.npz"))
    value, loc, attn, gout = [t.numpy() for t in full_shape_inputs()]
    chk = [float(a.astype(np.float64).sum()) for a in (value, loc, attn, gout)]
    np.testing.assert_allclose(chk, golden["input_checksum"], rtol=1e-12)      # same inputs as the generator saw
    shapes = np.asarray(FULL_LEVELS, np.int64)
    out = oracle.msda_forward(value, shapes, loc, attn)
    gv, gl, ga = oracle.msda_backward(value, shapes, loc, attn, gout)
    check_full_shape(out, gv, gl, ga, loc, golden)


def test_module_d32_fixture_inputs_regenerate():
    """tests/golden/msda_module_d32.npz stores only outputs; weights and inputs are regenerated from a seed by
    oracle/gen_golden.py:module_d32_inputs.  Their checksums must match what the fixture was generated from."""
    from conftest import Golden
    from oracle.gen_golden import MODULE_D32_CASES, module_d32_inputs
    g = Golden("msda_module_d32.npz")
    for name, refdim, use_mask, encoder in MODULE_D32_CASES:
        levels, sd, query, src, ref, mask, gout = module_d32_inputs(name, refdim, use_mask, encoder)
        got = [t.double().sum().item() for t in (query, src, ref, gout)] + [sd[k].double().sum().item() for k in sorted(sd)]
        np.testing.assert_allclose(got, g[name]["input_checksum"], rtol=1e-12)
        assert g[name]["out"].shape == (2, query.shape[1], 256)


def test_parallel_oracle_backward_is_bit_identical_to_serial():
    rng = np.random.default_rng(9)
    shapes = np.asarray([(9, 13), (5, 7), (3, 4)], np.int64)
    N, M, D, Lq, L, P = 2, 4, 32, 57, 3, 4
    S = int((shapes[:, 0] * shapes[:, 1]).sum())
    value = rng.random((N, S, M, D)).astype(np.float32)
    loc = (rng.random((N, Lq, M, L, P, 2)) * 1.3 - 0.15).astype(np.float32)
    attn = rng.random((N, Lq, M, L, P)).astype(np.float32)
    gout = rng.random((N, Lq, M * D)).astype(np.float32)
    a = oracle.msda_backward(value, shapes, loc, attn, gout)
    b = oracle.msda_backward(value, shapes, loc, attn, gout, parallel=True)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
