"""CPU: matcher / EMA / pseudo-label oracle against the reference-generated fixtures and scipy."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment

import oracle
from conftest import Golden

COST_CASES = Golden("cost.npz").names()
LSAP_CASES = Golden("lsap.npz").names()


def test_ioucost_docstring_known_answer(golden_cost):
    # thirdparty/mmdetection/mmdet/core/bbox/match_costs/match_cost.py:156-162
    want = np.array([[-0.1250, 0.1667], [0.1667, -0.5000]], np.float32)
    np.testing.assert_allclose(golden_cost.z["doc_ioucost"], want, atol=1e-4)
    b = np.array([[1, 1, 2, 2], [2, 2, 3, 4]], np.float32)
    cxcywh = np.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], -1)
    gt = np.array([[0, 0, 2, 4], [1, 2, 3, 4]], np.float32)
    _, _, iou, _ = (lambda t: (t[0], t[1], t[3], t[2]))(
        oracle.match_cost(cxcywh, np.zeros((2, 1), np.float32), gt, np.zeros(2, np.int64), 1.0, 1.0,
                          w_cls=0.0, w_reg=0.0, w_iou=1.0, parts=True))
    np.testing.assert_allclose(iou, want, atol=1e-4)


@pytest.mark.parametrize("case", COST_CASES)
def test_oracle_cost_matches_reference(case, golden_cost):
    g = golden_cost[case]
    if g["gt_bboxes"].shape[0] == 0:
        pytest.skip("no ground truth: the reference returns before building a matrix")
    w, h = g["img_wh"]
    total, cls, reg, iou = oracle.match_cost(g["bbox_pred"], g["cls_pred"], g["gt_bboxes"], g["gt_labels"],
                                             w, h, parts=True)
    np.testing.assert_allclose(total, g["cost"], rtol=1e-5, atol=1e-5)
    if "cost_cls" in g:
        np.testing.assert_allclose(cls, g["cost_cls"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(reg, g["cost_reg"], rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(iou, g["cost_iou"], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("case", COST_CASES)
def test_oracle_assignment_matches_reference(case, golden_cost):
    g = golden_cost[case]
    w, h = g["img_wh"]
    gi, lab, rows, cols = oracle.hungarian_assign(g["bbox_pred"], g["cls_pred"], g["gt_bboxes"],
                                                  g["gt_labels"], w, h)
    assert np.array_equal(rows, g["rows"]) and np.array_equal(cols, g["cols"])
    assert np.array_equal(gi, g["assigned_gt_inds"]) and np.array_equal(lab, g["assigned_labels"])


@pytest.mark.parametrize("case", LSAP_CASES)
def test_oracle_lsap_golden(case, golden_lsap):
    g = golden_lsap[case]
    rows, cols = oracle.lsap(g["cost"])
    assert np.array_equal(rows, g["rows"]) and np.array_equal(cols, g["cols"])


def test_oracle_lsap_random_vs_scipy():
    rng = np.random.default_rng(123)
    for t in range(400):
        nr, nc = int(rng.integers(1, 50)), int(rng.integers(1, 50))
        kind = t % 4
        if kind == 0:
            c = rng.random((nr, nc))
        elif kind == 1:
            c = rng.integers(0, 3, (nr, nc)).astype(float)
        elif kind == 2:
            c = np.round(rng.random((nr, nc)) * 4) / 4
            c[rng.random((nr, nc)) < 0.1] = np.inf
        else:
            c = rng.standard_normal((nr, nc)).astype(np.float32)
        try:
            want = linear_sum_assignment(c)
        except ValueError:
            with pytest.raises(oracle.LsapError):
                oracle.lsap(c)
            continue
        got = oracle.lsap(c)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_oracle_lsap_errors_and_empty():
    with pytest.raises(oracle.LsapError, match="invalid numeric"):
        oracle.lsap(np.array([[np.nan, 1.0]]))
    with pytest.raises(oracle.LsapError, match="invalid numeric"):
        oracle.lsap(np.array([[-np.inf, 1.0]]))
    with pytest.raises(oracle.LsapError, match="infeasible"):
        oracle.lsap(np.full((2, 2), np.inf))
    r, c = oracle.lsap(np.zeros((0, 3)))
    assert len(r) == 0 and len(c) == 0


def test_oracle_ema(golden_ema):
    z = golden_ema.z
    for wu in (0, 100):
        got = [oracle.ema_momentum(0.999, wu, int(s)) for s in z["sched_steps"]]
        assert np.array_equal(np.asarray(got), z[f"sched_wu{wu}"])
    worst = 0
    for mi in range(4):
        mom = float(z[f"m{mi}.momentum"])
        for si in range(6):
            t = z[f"m{mi}.t{si}.teacher"].copy()
            oracle.ema_update(t, z[f"m{mi}.t{si}.student"], mom)
            want = z[f"m{mi}.t{si}.out"]
            ulp = np.abs(t.view(np.int32).astype(np.int64) - want.view(np.int32).astype(np.int64)).max()
            worst = max(worst, int(ulp))
    assert worst <= 1, f"EMA oracle differs from torch by {worst} ulp"


def test_mean_teacher_host_logic_vs_reference_hook_sequences(golden_ema, monkeypatch):
    """The hook's host side (schedule :46-48, interval, initial clone :32-35, decay :52-58, wrapper unwrapping :27-28,
    positional parameter pairing incl. frozen parameters, buffers untouched) replayed against sequences the REFERENCE's own
    MeanTeacher produced; the device kernel is replaced by the oracle's update here (the GPU test runs the real one)."""
    import torch
    from conftest import drive_mean_teacher_sequence
    import semi_detr_amd.mean_teacher as mt
    monkeypatch.setattr(mt, "is_module_wrapper", lambda m: hasattr(m, "module") and not hasattr(m, "teacher"))

    def cpu_update(model, mom):
        for (_, ps), (_, pt) in zip(model.student.named_parameters(), model.teacher.named_parameters()):
            t = pt.detach().numpy()
            oracle.ema_update(t, ps.detach().numpy(), float(mom))

    z = golden_ema.z
    for name in z["seq.names"]:
        steps = 0
        for it, logged, hook_mom, teachers, model in drive_mean_teacher_sequence(z, str(name), "cpu", cpu_update):
            want = z[f"seq.{name}.logged_momentum"][it]
            assert (np.isnan(want) and np.isnan(logged)) or logged == want, (name, it, logged, want)
            assert hook_mom == z[f"seq.{name}.hook_momentum"][it], (name, it)
            for i, t in enumerate(teachers):
                ref = z[f"seq.{name}.teacher{it + 1}.{i}"]
                ulp = np.abs(t.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64)).max()
                assert ulp <= 2, (name, it, i, int(ulp))       # oracle rounding vs torch CPU: <= 1 ulp per update
            steps += 1
        assert steps == int(z["seq.iters"])
        assert np.array_equal(model.teacher.buf.numpy(), z[f"seq.{name}.buf_end"])       # buffers are never averaged


def test_oracle_pseudo(golden_pseudo):
    for name in golden_pseudo.names():
        g = golden_pseudo[name]
        keep, thr = oracle.pseudo_label_filter(g["proposal"])
        assert np.array_equal(keep, g["keep"]), name
        if g["proposal"].shape[0] > 1:
            np.testing.assert_allclose(thr, g["thr"], rtol=2e-7)
