"""GPU parity of the fused task-aligned focal loss (semidetr_tal_loss_f32) against the reference's own function
(values + autograd gradients in tests/golden/tal_loss.npz) and the C oracle; both entry contracts: probabilities
(the reference module's forward) and raw logits (sigmoid fused)."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
TAL = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tal_loss.npz"))


def case(name):
    return {k.split(".", 1)[1]: TAL[k] for k in TAL.files if k.startswith(name + ".")}


@pytest.mark.parametrize("name", list(TAL["names"]))
def test_tal_loss_matches_reference_fixture(name):
    from semi_detr_amd import TaskAlignedFocalLoss
    g = case(name)
    avg = float(g["avg_factor"])
    crit = TaskAlignedFocalLoss(use_sigmoid=True, gamma=2.0, loss_weight=1.0)
    lab, met = torch.from_numpy(g["labels"]).cuda(), torch.from_numpy(g["metric"]).cuda()
    # (a) fused with the sigmoid
    x = torch.from_numpy(g["logits"]).cuda().requires_grad_(True)
    loss = crit.forward_logits(x, lab, met, avg_factor=avg)
    loss.backward()
    ref = float(g["loss"])
    assert abs(float(loss.detach()) - ref) <= 3e-6 * max(abs(ref), 1e-3)
    scale = np.abs(g["grad_logits"]).max() + 1e-12
    np.testing.assert_allclose(x.grad.cpu().numpy(), g["grad_logits"], rtol=3e-5, atol=3e-6 * scale)
    # (b) the module's own contract: probabilities in, autograd continues through the caller's sigmoid
    x2 = torch.from_numpy(g["logits"]).cuda().requires_grad_(True)
    loss2 = crit(x2.sigmoid(), lab, met, avg_factor=avg)
    loss2.backward()
    assert abs(float(loss2.detach()) - ref) <= 3e-6 * max(abs(ref), 1e-3)
    ok = np.abs(g["grad_logits"]) > 1e-3 * scale             # p(1-p) underflows where the sigmoid saturates
    np.testing.assert_allclose(x2.grad.cpu().numpy()[ok], g["grad_logits"][ok], rtol=2e-3, atol=3e-6 * scale)
    # (c) oracle: same fp32 formulas, transcendental ulps apart; sum within fp32 summation error
    s, grad = oracle.tal_loss(g["logits"], g["labels"], g["metric"], input_is_prob=False)
    np.testing.assert_allclose(x.grad.cpu().numpy() * avg, grad, rtol=2e-5, atol=1e-7 * scale * avg)
    assert abs(float(loss.detach()) * avg - s) <= 2e-6 * max(abs(s), 1e-3)


def test_tal_loss_large_and_reductions():
    from semi_detr_amd import TaskAlignedFocalLoss, task_aligned_focal_loss
    rng = np.random.default_rng(2)
    N, C = 7 * 5 * 900, 80                                    # all decoder layers x images of one loss() call
    logits = rng.normal(-2, 2, (N, C)).astype(np.float32)
    lab = np.full(N, C, np.int64)
    pos = rng.choice(N, N // 10, replace=False)
    lab[pos] = rng.integers(0, C, len(pos))
    met = np.zeros(N, np.float32)
    met[pos] = rng.random(len(pos))
    x = torch.from_numpy(logits).cuda().requires_grad_(True)
    tl, tm = torch.from_numpy(lab).cuda(), torch.from_numpy(met).cuda()
    total = task_aligned_focal_loss(x, tl, tm, reduction="sum", from_logits=True)
    (total * 0.5).backward()
    s, grad = oracle.tal_loss(logits, lab, met, input_is_prob=False)
    assert abs(float(total.detach()) - s) <= 3e-6 * s
    np.testing.assert_allclose(x.grad.cpu().numpy(), 0.5 * grad, rtol=2e-5, atol=1e-9)      # expf/logf: libm vs ocml ulps
    again = task_aligned_focal_loss(x.detach(), tl, tm, reduction="sum", from_logits=True)
    assert float(again) == float(total.detach())                       # deterministic summation
    mean = task_aligned_focal_loss(x.detach(), tl, tm, reduction="mean", from_logits=True)
    assert abs(float(mean) - s / (N * C)) <= 3e-6 * s / (N * C)
    crit = TaskAlignedFocalLoss(loss_weight=2.0, gamma=1.5)
    l15 = crit.forward_logits(x.detach()[:1000], tl[:1000], tm[:1000], avg_factor=3.0)
    s15, _ = oracle.tal_loss(logits[:1000], lab[:1000], met[:1000], gamma=1.5, input_is_prob=False, want_grad=False)
    assert abs(float(l15) - 2.0 * s15 / 3.0) <= 2e-5 * abs(2.0 * s15 / 3.0)


def test_tal_loss_errors():
    from semi_detr_amd import TaskAlignedFocalLoss, task_aligned_focal_loss
    x, t, m = torch.rand(4, 3).cuda(), torch.tensor([0, 3, 1, 3]).cuda(), torch.rand(4).cuda()
    with pytest.raises(NotImplementedError):
        task_aligned_focal_loss(x, t, m, weight=torch.ones(4).cuda())
    with pytest.raises(NotImplementedError):
        task_aligned_focal_loss(x, t, m, reduction="none")
    with pytest.raises(ValueError):
        task_aligned_focal_loss(x, t, m, reduction="sum", avg_factor=2.0)
    with pytest.raises(ValueError):
        task_aligned_focal_loss(x, t[:3], m)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        TaskAlignedFocalLoss()(x.cpu(), t.cpu(), m.cpu())
    assert float(task_aligned_focal_loss(torch.zeros(0, 3).cuda(), t[:0], m[:0], reduction="sum")) == 0.0
