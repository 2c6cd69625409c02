"""CPU: the NMS / box-warp oracle (oracle/nms_oracle.c) against fixtures made from the reference's own Python
(tests/golden/nms.npz, transform.npz; oracle/gen_golden.py).  The fixtures' greedy suppression is a torch
restatement of the un-vendored mmcv-full 1.3.16 ops (parity unpinned for the keep decisions, see the oracle's
header); everything around it -- decoding, thresholds, multiclass_nms, top-k, Transform2D -- is reference code."""
import os

import numpy as np
import pytest

import oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name))


NMS = _load("nms.npz")
TRF = _load("transform.npz")


@pytest.mark.parametrize("case", list(NMS["names"]))
def test_nms_oracle_matches_reference_fixture(case):
    g = {k.split(".", 1)[1]: NMS[k] for k in NMS.files if k.startswith(case + ".")}
    dets, labels = oracle.pseudo_nms(g["logits"], g["bbox_pred"], g["img_hw"][0], g["img_hw"][1],
                                     max_num=int(g["max_per_img"]))
    assert dets.shape == g["dets"].shape, (dets.shape, g["dets"].shape)
    np.testing.assert_array_equal(labels, g["labels"])
    np.testing.assert_array_equal(dets[:, :4], g["dets"][:, :4])            # boxes: same fp32 operations
    np.testing.assert_allclose(dets[:, 4], g["dets"][:, 4], rtol=0, atol=2e-7)   # sigmoid: libm vs torch


def test_nms_fixture_is_not_trivial():
    g = NMS
    assert len(g["dino.dets"]) == 300 and len(g["none.dets"]) == 0 and len(g["topk.dets"]) == 50
    # suppression really happened in the clustered cases: fewer survivors than candidates above the threshold
    s = 1 / (1 + np.exp(-g["few.logits"].astype(np.float64)))
    assert 0 < len(g["few.dets"]) < int((s > 0.01).sum())


def test_nms_oracle_properties():
    rng = np.random.default_rng(3)
    Q, C = 90, 6
    logits = np.round(rng.normal(-1, 2, (Q, C)) * 32) / 32
    bp = np.concatenate([rng.random((Q, 2)), rng.random((Q, 2)) * 0.4 + 0.05], -1)
    dets, labels = oracle.pseudo_nms(logits, bp, 480, 640, max_num=0)
    # sorted by score, no two kept boxes of one class overlap by more than the threshold
    assert np.all(np.diff(dets[:, 4]) <= 0)
    for c in range(C):
        b = dets[labels == c, :4].astype(np.float64)
        for i in range(len(b)):
            for j in range(i + 1, len(b)):
                w = max(min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0]), 0)
                h = max(min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1]), 0)
                a = lambda t: (t[2] - t[0]) * (t[3] - t[1])
                assert w * h / (a(b[i]) + a(b[j]) - w * h) <= 0.6 + 1e-5
    # idempotent: NMS of the survivors (as one-hot logits) keeps them all
    dets1, _ = oracle.pseudo_nms(logits, bp, 480, 640, iou_thr=1.0, max_num=0)
    assert len(dets1) == int((1 / (1 + np.exp(-logits.astype(np.float32))) > 0.01).sum())
    # empty inputs
    d0, l0 = oracle.pseudo_nms(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32), 10, 10)
    assert d0.shape == (0, 5) and l0.shape == (0,)


@pytest.mark.parametrize("case", list(TRF["names"]))
def test_transform_oracle_matches_reference_fixture(case):
    g = {k.split(".", 1)[1]: TRF[k] for k in TRF.files if k.startswith(case + ".")}
    out = oracle.transform_bboxes(g["boxes"][:, :4], g["M"], g["out_shape"][0], g["out_shape"][1])
    assert out.shape == g["out"][:, :4].shape
    # torch.matmul's accumulation order is BLAS-defined: a few ulps of ~1000-pixel coordinates
    np.testing.assert_allclose(out, g["out"][:, :4], rtol=0, atol=5e-4)
    if g["out"].shape[1] == 5:
        np.testing.assert_array_equal(g["out"][:, 4], g["boxes"][:, 4])
