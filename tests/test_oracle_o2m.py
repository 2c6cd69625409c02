"""CPU: the one-to-many assigner oracle (oracle/o2m_oracle.c) against fixtures produced by the reference's own
O2MAssigner (detr_od/core/bbox/assigners/o2m_assigner.py, imported by path in oracle/gen_golden.py) and by the
warm-up branch of _get_target_single restated with the same torch calls (tests/golden/o2m.npz)."""
import os

import numpy as np
import pytest

import oracle

O2M = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "o2m.npz"))


def case(name):
    return {k.split(".", 1)[1]: O2M[k] for k in O2M.files if k.startswith(name + ".")}


@pytest.mark.parametrize("name", list(O2M["names"]))
def test_o2m_oracle_matches_reference_fixture(name):
    g = case(name)
    ih, iw = g["img_hw"]
    gi, lab, mo, am = oracle.o2m_assign(g["bbox_pred"], g["cls_prob"], g["gt_bboxes"], g["gt_labels"], iw, ih)
    np.testing.assert_array_equal(gi, g["gt_inds"])
    np.testing.assert_array_equal(lab, g["labels"])
    np.testing.assert_allclose(mo, g["max_overlaps"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(am, g["assign_metrics"], rtol=5e-6, atol=1e-9)       # x**6: torch.pow vs squaring
    C = g["cls_prob"].shape[1]
    lf, bt, nm = oracle.o2m_targets(gi, mo, am, g["gt_bboxes"], g["gt_labels"], iw, ih, C)
    np.testing.assert_array_equal(lf, g["labels_full"])
    np.testing.assert_allclose(bt, g["bbox_targets"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(nm, g["norm_metrics"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("name", list(O2M["names"]))
@pytest.mark.parametrize("tag,kw", [("t1", dict(topk=1)), ("tk", dict(topk=13, dynamic_k=True))])
def test_o2m_oracle_teacher_options_match_reference_fixture(name, tag, kw):
    """teacher_assign (best-aligned candidate only, o2m_assigner.py:115-119) and teacher_assign + multiple_pos (dynamic k,
    :125-133), both produced by the reference's own O2MAssigner.assign."""
    g = case(name)
    ih, iw = g["img_hw"]
    gi, lab, mo, am = oracle.o2m_assign(g["bbox_pred"], g["cls_prob"], g["gt_bboxes"], g["gt_labels"], iw, ih, **kw)
    np.testing.assert_array_equal(gi, g[f"{tag}_gt_inds"])
    np.testing.assert_array_equal(lab, g[f"{tag}_labels"])
    np.testing.assert_allclose(mo, g[f"{tag}_max_overlaps"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(am, g[f"{tag}_assign_metrics"], rtol=5e-6, atol=1e-9)


def test_o2m_fixture_is_not_trivial():
    g = case("dup_gt")
    # overlapping ground truths: some query is a top-13 candidate of two gts and goes to the one with the larger IoU
    assert (g["gt_inds"] > 0).sum() < 13 * len(g["gt_labels"])
    assert case("no_gt")["gt_inds"].tolist() == [0] * 50 and (case("no_gt")["labels"] == -1).all()
    assert (case("one_gt")["gt_inds"] > 0).sum() == 13


def test_o2m_oracle_properties():
    rng = np.random.default_rng(1)
    Q, C, G = 120, 7, 5
    gt = np.concatenate([rng.random((G, 2)) * 300, rng.random((G, 2)) * 200 + 320], -1).astype(np.float32)
    bp = np.concatenate([rng.random((Q, 2)), rng.random((Q, 2)) * 0.5 + 0.1], -1).astype(np.float32)
    prob = rng.random((Q, C)).astype(np.float32)
    gl = rng.integers(0, C, G)
    gi, lab, mo, am = oracle.o2m_assign(bp, prob, gt, gl, 640, 480, topk=4)
    for g_ in range(G):
        assert (gi == g_ + 1).sum() <= 4                         # at most top-k positives per gt
    assert ((gi > 0) == (mo != -1e8)).all() and (am[gi == 0] == 0).all()
    assert (lab[gi > 0] == gl[gi[gi > 0] - 1]).all() and (lab[gi == 0] == -1).all()
    # top-1 == teacher_assign option 1
    gi1, _, _, _ = oracle.o2m_assign(bp, prob, gt, gl, 640, 480, topk=1)
    assert (gi1 > 0).sum() <= G
