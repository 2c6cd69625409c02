"""CPU: the task-aligned focal loss oracle (oracle/o2m_oracle.c:tal_loss_oracle) against values and autograd
gradients of the reference's own task_aigned_focal_loss (tests/golden/tal_loss.npz, oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

import oracle

TAL = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tal_loss.npz"))


def case(name):
    return {k.split(".", 1)[1]: TAL[k] for k in TAL.files if k.startswith(name + ".")}


@pytest.mark.parametrize("name", list(TAL["names"]))
def test_tal_oracle_matches_reference_fixture(name):
    g = case(name)
    avg = float(g["avg_factor"])
    s, grad = oracle.tal_loss(g["logits"], g["labels"], g["metric"], input_is_prob=False)
    assert abs(s / avg - float(g["loss"])) <= 2e-6 * max(abs(float(g["loss"])), 1e-3)
    scale = np.abs(g["grad_logits"]).max() + 1e-12
    np.testing.assert_allclose(grad / avg, g["grad_logits"], rtol=2e-5, atol=2e-6 * scale)
    prob = (1.0 / (1.0 + np.exp(-g["logits"].astype(np.float32)))).astype(np.float32)
    s2, gp = oracle.tal_loss(prob, g["labels"], g["metric"], input_is_prob=True)
    assert abs(s2 - s) <= 1e-5 * max(abs(s), 1e-3)
    # saturated probabilities (p == 1 in fp32) make BCE's own gradient huge: compare where the reference's is finite
    ok = np.isfinite(g["grad_prob"]) & (np.abs(g["grad_prob"]) < 1e6)
    np.testing.assert_allclose((gp / avg)[ok], g["grad_prob"][ok], rtol=3e-4, atol=1e-6)


def test_tal_oracle_finite_differences():
    rng = np.random.default_rng(0)
    N, C = 9, 4
    x = rng.normal(0, 1.5, (N, C)).astype(np.float32)
    lab = rng.integers(0, C + 1, N)
    met = (rng.random(N) * (lab < C)).astype(np.float32)
    for gamma in (2.0, 1.5):
        s, grad = oracle.tal_loss(x, lab, met, gamma=gamma, input_is_prob=False)
        for (i, c) in ((0, 0), (3, 2), (8, 3)):
            xp, xm = x.copy(), x.copy()
            xp[i, c] += 1e-2; xm[i, c] -= 1e-2
            fd = (oracle.tal_loss(xp, lab, met, gamma=gamma, input_is_prob=False, want_grad=False)[0] -
                  oracle.tal_loss(xm, lab, met, gamma=gamma, input_is_prob=False, want_grad=False)[0]) / 2e-2
            assert abs(fd - grad[i, c]) < 2e-3 * max(1.0, abs(fd))
