"""GPU parity of the teacher box decoding + class-aware NMS + box warp through the
C ABI (semidetr_pseudo_nms_f32, semidetr_transform_bboxes_f32, semidetr_pseudo_label_filter_f32 with counts):
against the fixtures made from the reference's Python and against the CPU oracle on seeded inputs.
Labels / kept sets / order are exact, boxes are bit-exact (same fp32 operations), scores within 2e-7 (expf)."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NMS = np.load(os.path.join(GOLD, "nms.npz"))
TRF = np.load(os.path.join(GOLD, "transform.npz"))


def _meta(h, w):
    return dict(img_shape=(int(h), int(w), 3), scale_factor=np.ones(4, np.float32))


def _check(res, exp_dets, exp_labels):
    dets, labels = res
    assert dets.shape == exp_dets.shape, (dets.shape, exp_dets.shape)
    np.testing.assert_array_equal(labels.cpu().numpy(), exp_labels)
    np.testing.assert_array_equal(dets[:, :4].cpu().numpy(), exp_dets[:, :4])
    np.testing.assert_allclose(dets[:, 4].cpu().numpy(), exp_dets[:, 4], rtol=0, atol=2e-7)


@pytest.mark.parametrize("case", list(NMS["names"]))
def test_nms_matches_reference_fixture(case):
    from semi_detr_amd import get_bboxes_for_pseudo_label
    g = {k.split(".", 1)[1]: NMS[k] for k in NMS.files if k.startswith(case + ".")}
    res = get_bboxes_for_pseudo_label(torch.from_numpy(g["logits"])[None].cuda(), torch.from_numpy(g["bbox_pred"])[None].cuda(),
                                      [_meta(*g["img_hw"])], max_per_img=int(g["max_per_img"]))
    assert len(res) == 1
    _check(res[0], g["dets"], g["labels"])


def _random_batch(seed, B, Q, C, bias, quant=None, spread=2.0):
    rng = np.random.default_rng(seed)
    logits = rng.normal(bias, spread, (B, Q, C)).astype(np.float32)
    if quant:
        logits = (np.round(logits * quant) / quant).astype(np.float32)
    k = max(Q // 8, 1)
    cxcy = rng.random((B, Q, 2))
    wh = rng.random((B, Q, 2)) * 0.3 + 0.02
    for b in range(B):           # clusters of near duplicates
        src = rng.integers(0, k, Q - k)
        cxcy[b, k:] = cxcy[b, src] + rng.normal(0, 0.01, (Q - k, 2))
        wh[b, k:] = wh[b, src] * (1 + rng.normal(0, 0.05, (Q - k, 2)))
    bbox = np.concatenate([cxcy, wh], -1).astype(np.float32)
    shapes = [(int(rng.integers(400, 900)), int(rng.integers(500, 1400))) for _ in range(B)]
    return logits, bbox, shapes


@pytest.mark.parametrize("name,B,Q,C,bias,quant,max_per_img", [
    ("dino_teacher", 5, 900, 80, -5.0, None, 300),       # the SSOD batch: sparse confident scores
    ("all_candidates", 2, 900, 80, 2.0, None, 300),      # every (query, class) above the threshold: radix select
    ("all_candidates_cap", 1, 300, 80, 2.0, None, 2048),
    ("ties", 3, 200, 12, -1.0, 4, 100),                  # many exactly equal logits: the tie rule
    ("tiny", 4, 3, 2, 0.0, None, 300),
    ("wide", 2, 1500, 3, -2.0, None, 300),               # Q > 1024: the 2048-slot sort
    ("max_queries", 1, 2048, 2, 1.0, None, 2048),        # the size limits of the entry point
    ("many_classes", 2, 100, 365, -3.0, None, 300),      # Objects365-sized class count
])
def test_nms_batch_vs_oracle(name, B, Q, C, bias, quant, max_per_img):
    from semi_detr_amd import get_bboxes_for_pseudo_label
    logits, bbox, shapes = _random_batch(len(name) * 7 + Q, B, Q, C, bias, quant)
    res = get_bboxes_for_pseudo_label(torch.from_numpy(logits).cuda(), torch.from_numpy(bbox).cuda(),
                                      [_meta(h, w) for h, w in shapes], max_per_img=max_per_img)
    assert len(res) == B
    for b in range(B):
        exp = oracle.pseudo_nms(logits[b], bbox[b], shapes[b][0], shapes[b][1], max_num=max_per_img)
        _check(res[b], *exp)


def test_nms_thresholds_are_arguments():
    from semi_detr_amd import get_bboxes_for_pseudo_label
    logits, bbox, shapes = _random_batch(5, 2, 300, 20, -2.0)
    for thr, iou in ((0.3, 0.3), (0.05, 0.9)):
        res = get_bboxes_for_pseudo_label(torch.from_numpy(logits).cuda(), torch.from_numpy(bbox).cuda(),
                                          [_meta(h, w) for h, w in shapes], score_thr=thr, iou_threshold=iou,
                                          max_per_img=150)
        for b in range(2):
            _check(res[b], *oracle.pseudo_nms(logits[b], bbox[b], *shapes[b], score_thr=thr, iou_thr=iou, max_num=150))


def test_teacher_pseudo_labels_chain_equals_steps_and_oracle():
    """NMS -> mean+std filter chained on the device == the two steps run separately == the oracle's composition."""
    from semi_detr_amd import filter_pseudo_labels, get_bboxes_for_pseudo_label, teacher_pseudo_labels
    logits, bbox, shapes = _random_batch(11, 5, 900, 80, -5.0)
    metas = [_meta(h, w) for h, w in shapes]
    tl, tb = torch.from_numpy(logits).cuda(), torch.from_numpy(bbox).cuda()
    boxes, labels, scores, props = teacher_pseudo_labels(tl, tb, metas, return_proposals=True)
    pending = teacher_pseudo_labels(tl, tb, metas, wait=False)          # read-back queued; result() waits for its event
    later = pending.result(return_proposals=True)
    assert all(torch.equal(a, b) for x, y in zip(later[:3], (boxes, labels, scores)) for a, b in zip(x, y))
    assert pending.result() is not None and len(pending.result()) == 3
    sep = get_bboxes_for_pseudo_label(tl, tb, metas)
    b2, l2, s2 = filter_pseudo_labels([p[0] for p in sep], [p[1] for p in sep])
    for b in range(5):
        assert torch.equal(props[b][0], sep[b][0]) and torch.equal(props[b][1], sep[b][1])
        assert torch.equal(boxes[b], b2[b]) and torch.equal(labels[b], l2[b]) and torch.equal(scores[b], s2[b])
        dets, labs = oracle.pseudo_nms(logits[b], bbox[b], *shapes[b])
        keep, _ = oracle.pseudo_label_filter(dets)
        np.testing.assert_array_equal(boxes[b].cpu().numpy(), dets[keep, :4])
        np.testing.assert_array_equal(labels[b].cpu().numpy(), labs[keep])
        assert 0 < len(keep) < len(dets)


@pytest.mark.parametrize("case", list(TRF["names"]))
def test_transform_matches_reference_fixture_and_oracle(case):
    from semi_detr_amd import transform_bboxes
    g = {k.split(".", 1)[1]: TRF[k] for k in TRF.files if k.startswith(case + ".")}
    box = torch.from_numpy(g["boxes"]).cuda()
    out = transform_bboxes(box, torch.from_numpy(g["M"]).cuda(), tuple(g["out_shape"]))
    assert out.shape == g["out"].shape
    np.testing.assert_allclose(out.cpu().numpy()[:, :4], g["out"][:, :4], rtol=0, atol=5e-4)
    if len(g["boxes"]):
        np.testing.assert_array_equal(out.cpu().numpy()[:, 4], g["boxes"][:, 4])
        exp = oracle.transform_bboxes(g["boxes"][:, :4], g["M"], *g["out_shape"])
        np.testing.assert_allclose(out.cpu().numpy()[:, :4], exp, rtol=0, atol=2e-4)


def test_transform_list_form_one_launch():
    from semi_detr_amd import transform_bboxes
    cases = [{k.split(".", 1)[1]: TRF[k] for k in TRF.files if k.startswith(n + ".")} for n in TRF["names"]]
    outs = transform_bboxes([torch.from_numpy(g["boxes"][:, :4].copy()).cuda() for g in cases],
                            [torch.from_numpy(g["M"]) for g in cases], [tuple(g["out_shape"]) for g in cases])
    for g, o in zip(cases, outs):
        assert o.shape == (len(g["boxes"]), 4)
        np.testing.assert_allclose(o.cpu().numpy(), g["out"][:, :4], rtol=0, atol=5e-4)
    assert transform_bboxes([], [], []) == []


def test_nms_errors():
    from semi_detr_amd import get_bboxes_for_pseudo_label, transform_bboxes
    l, b = torch.zeros(1, 4, 3).cuda(), torch.zeros(1, 4, 4).cuda()
    with pytest.raises(RuntimeError, match="max_per_img"):
        get_bboxes_for_pseudo_label(l, b, [_meta(10, 10)], max_per_img=0)
    with pytest.raises(RuntimeError, match="2048 queries"):
        get_bboxes_for_pseudo_label(torch.zeros(1, 2049, 1).cuda(), torch.zeros(1, 2049, 4).cuda(), [_meta(10, 10)])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        get_bboxes_for_pseudo_label(l.cpu(), b.cpu(), [_meta(10, 10)])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        transform_bboxes(torch.zeros(2, 4), torch.eye(3), (5, 5))
    with pytest.raises(ValueError):
        get_bboxes_for_pseudo_label(l, torch.zeros(1, 5, 4).cuda(), [_meta(10, 10)])
    res = get_bboxes_for_pseudo_label(torch.zeros(2, 0, 3).cuda(), torch.zeros(2, 0, 4).cuda(), [_meta(10, 10)] * 2)
    assert [tuple(r[0].shape) for r in res] == [(0, 5), (0, 5)]
