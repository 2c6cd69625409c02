"""GPU parity of the one-to-many assigner + warm-up targets through the C ABI
(semidetr_o2m_assign_f32): against fixtures produced by the reference's own O2MAssigner and against the C oracle.
Indices / labels exact; IoUs, metrics and targets within a few fp32 ulps (bit-exact against the oracle)."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
O2M = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "o2m.npz"))


def case(name):
    return {k.split(".", 1)[1]: O2M[k] for k in O2M.files if k.startswith(name + ".")}


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name", list(O2M["names"]))
def test_o2m_assign_matches_reference_fixture(name):
    from semi_detr_amd import O2MAssigner
    g = case(name)
    ih, iw = (int(v) for v in g["img_hw"])
    res = O2MAssigner().assign(_t(g["bbox_pred"]), _t(g["cls_prob"]), _t(g["gt_bboxes"]), _t(g["gt_labels"]),
                               dict(img_shape=(ih, iw, 3)))
    assert res.num_gts == len(g["gt_labels"]) and res.num_preds == len(g["gt_inds"])
    np.testing.assert_array_equal(res.gt_inds.cpu().numpy(), g["gt_inds"])
    np.testing.assert_array_equal(res.labels.cpu().numpy(), g["labels"])
    np.testing.assert_allclose(res.max_overlaps.cpu().numpy(), g["max_overlaps"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(res.assign_metrics.cpu().numpy(), g["assign_metrics"], rtol=5e-6, atol=1e-9)


@pytest.mark.parametrize("name", list(O2M["names"]))
@pytest.mark.parametrize("tag,kw", [("t1", dict(teacher_assign=True)), ("tk", dict(teacher_assign=True, multiple_pos=True))])
def test_o2m_teacher_options_match_reference_fixture(name, tag, kw):
    """The teacher's two options of O2MAssigner.assign (o2m_assigner.py:115-133): best-aligned candidate only, and the
    dynamic-k `multiple_pos` branch -- outputs of the reference's own assign()."""
    from semi_detr_amd import O2MAssigner
    g = case(name)
    ih, iw = (int(v) for v in g["img_hw"])
    res = O2MAssigner().assign(_t(g["bbox_pred"]), _t(g["cls_prob"]), _t(g["gt_bboxes"]), _t(g["gt_labels"]),
                               dict(img_shape=(ih, iw, 3)), **kw)
    np.testing.assert_array_equal(res.gt_inds.cpu().numpy(), g[f"{tag}_gt_inds"])
    np.testing.assert_array_equal(res.labels.cpu().numpy(), g[f"{tag}_labels"])
    np.testing.assert_allclose(res.max_overlaps.cpu().numpy(), g[f"{tag}_max_overlaps"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(res.assign_metrics.cpu().numpy(), g[f"{tag}_assign_metrics"], rtol=5e-6, atol=1e-9)
    # and bit for bit against the oracle on a random problem with many zero-metric candidates (dynamic k keeps them)
    rng = np.random.default_rng(len(name))
    Q, C, G = 333, 6, 9
    gt = np.concatenate([rng.random((G, 2)) * 300, rng.random((G, 2)) * 200 + 320], -1).astype(np.float32)
    bp = np.concatenate([rng.random((Q, 2)), rng.random((Q, 2)) * 0.5 + 0.1], -1).astype(np.float32)
    prob = (rng.random((Q, C)) * (rng.random((Q, C)) > 0.5)).astype(np.float32)
    gl = rng.integers(0, C, G)
    r2 = O2MAssigner(candidate_topk=7).assign(_t(bp), _t(prob), _t(gt), _t(gl), dict(img_shape=(480, 640, 3)), **kw)
    okw = dict(topk=1) if tag == "t1" else dict(topk=7, dynamic_k=True)
    gi, lab, mo, am = oracle.o2m_assign(bp, prob, gt, gl, 640, 480, **okw)
    np.testing.assert_array_equal(r2.gt_inds.cpu().numpy(), gi)
    np.testing.assert_array_equal(r2.max_overlaps.cpu().numpy(), mo)
    np.testing.assert_array_equal(r2.assign_metrics.cpu().numpy(), am)


def test_o2m_batch_targets_match_reference_fixture_and_oracle():
    """All fixture cases with the same (Q, C) cannot be stacked (sizes differ), so: the DINO case replicated as a
    7-layer x 3-image batch with per-problem ground truths taken from different fixtures' boxes."""
    from semi_detr_amd import O2MAssigner
    g = case("dino")
    rng = np.random.default_rng(0)
    B, Q, C = 6, 900, 80
    bp = np.stack([np.roll(g["bbox_pred"], s, 0) for s in range(B)])
    cp = np.stack([np.roll(g["cls_prob"], 3 * s, 0) for s in range(B)])
    gts, gls, metas = [], [], []
    for b in range(B):
        G = [7, 0, 1, 30, 7, 100][b]
        src = rng.integers(0, 7, G)
        gt = g["gt_bboxes"][src] + rng.random((G, 4)).astype(np.float32) * 40
        gt[:, 2:] = np.maximum(gt[:, 2:], gt[:, :2] + 4)
        if b == 0:
            gt = g["gt_bboxes"].copy()
        gts.append(gt.astype(np.float32)); gls.append(rng.integers(0, C, G) if b else g["gt_labels"])
        metas.append(dict(img_shape=(800 - 10 * b, 1333 - 20 * b, 3)) if b else dict(img_shape=(800, 1333, 3)))
    out = O2MAssigner().assign_batch(_t(bp), _t(cp), [_t(x) for x in gts], [_t(np.asarray(x, np.int64)) for x in gls], metas)
    # problem 0 is the fixture itself (incl. the targets built by the reference's head code)
    np.testing.assert_array_equal(out["gt_inds"][0].cpu().numpy(), g["gt_inds"])
    np.testing.assert_array_equal(out["labels_full"][0].cpu().numpy(), g["labels_full"])
    np.testing.assert_allclose(out["bbox_targets"][0].cpu().numpy(), g["bbox_targets"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(out["norm_metrics"][0].cpu().numpy(), g["norm_metrics"], rtol=1e-5, atol=1e-8)
    for b in range(B):
        ih, iw = metas[b]["img_shape"][:2]
        gi, lab, mo, am = oracle.o2m_assign(bp[b], cp[b], gts[b], gls[b], iw, ih)
        lf, bt, nm = oracle.o2m_targets(gi, mo, am, gts[b], gls[b], iw, ih, C)
        for key, want in (("gt_inds", gi), ("labels", lab), ("max_overlaps", mo), ("assign_metrics", am),
                          ("labels_full", lf), ("bbox_targets", bt), ("norm_metrics", nm)):
            np.testing.assert_array_equal(out[key][b].cpu().numpy(), want, err_msg=f"{key}[{b}]")
    assert out["num_gts"] == [7, 0, 1, 30, 7, 100]


@pytest.mark.parametrize("Q,C,G,topk,alpha,beta", [(1500, 11, 9, 13, 1, 6), (64, 3, 5, 1, 1, 6), (300, 20, 12, 20, 0.5, 2.0),
                                                   (200, 4, 3, 200, 1, 6), (2048, 91, 300, 13, 1, 6), (900, 80, 1024, 13, 1, 6)])
def test_o2m_random_vs_oracle(Q, C, G, topk, alpha, beta):
    from semi_detr_amd import O2MAssigner
    rng = np.random.default_rng(Q + G)
    iw, ih = 640, 480
    gt = np.concatenate([rng.random((G, 2)) * [400, 300], np.zeros((G, 2))], -1)
    gt[:, 2:] = gt[:, :2] + rng.random((G, 2)) * 200 + 10
    gt = gt.astype(np.float32)
    f = np.asarray([iw, ih, iw, ih], np.float32)
    bp = np.concatenate([rng.random((Q, 2)), rng.random((Q, 2)) * 0.4 + 0.02], -1).astype(np.float32)
    src = rng.integers(0, G, Q // 2)
    n = gt[src] / f
    near = np.stack([(n[:, 0] + n[:, 2]) / 2, (n[:, 1] + n[:, 3]) / 2, n[:, 2] - n[:, 0], n[:, 3] - n[:, 1]], -1)
    bp[:Q // 2] = (near * (1 + rng.normal(0, 0.1, near.shape))).clip(0.001, 0.999)
    prob = (rng.random((Q, C)) ** 2).astype(np.float32)
    gl = rng.integers(0, C, G)
    out = O2MAssigner(candidate_topk=topk).assign_batch(_t(bp)[None], _t(prob)[None], [_t(gt)], [_t(gl)],
                                                        [dict(img_shape=(ih, iw, 3))], alpha=alpha, beta=beta)
    gi, lab, mo, am = oracle.o2m_assign(bp, prob, gt, gl, iw, ih, topk=topk, alpha=alpha, beta=beta)
    np.testing.assert_array_equal(out["gt_inds"][0].cpu().numpy(), gi)
    np.testing.assert_array_equal(out["labels"][0].cpu().numpy(), lab)
    exact = float(beta) in (1.0, 2.0, 6.0) and float(alpha) in (1.0, 2.0, 6.0)      # powf: libm vs ocml differ by ulps
    tol = dict(rtol=0, atol=0) if exact else dict(rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(out["max_overlaps"][0].cpu().numpy(), mo, rtol=0, atol=0)
    np.testing.assert_allclose(out["assign_metrics"][0].cpu().numpy(), am, **tol)
    assert (gi > 0).sum() > 0


def test_o2m_errors_and_teacher_assign():
    from semi_detr_amd import O2MAssigner
    a = O2MAssigner()
    meta = dict(img_shape=(100, 100, 3))
    bp, pr = torch.rand(8, 4).cuda(), torch.rand(8, 3).cuda()
    gt, gl = torch.tensor([[10.0, 10, 50, 50]]).cuda(), torch.tensor([1]).cuda()
    with pytest.raises(RuntimeError, match="selected index k out of range"):       # torch.topk's error, o2m_assigner.py:121
        a.assign(bp, pr, gt, gl, meta)
    r = a.assign(bp, pr, gt[:0], gl[:0], meta)                                      # no gts: fine even with Q < k
    assert r.gt_inds.tolist() == [0] * 8 and r.labels.tolist() == [-1] * 8 and r.max_overlaps.tolist() == [0.0] * 8
    bp2 = torch.cat([torch.tensor([[0.3, 0.3, 0.4, 0.4]]).cuda().repeat(20, 1) + torch.rand(20, 4).cuda() * 0.02, bp])
    pr2 = torch.rand(28, 3).cuda()
    r1 = a.assign(bp2, pr2, gt, gl, meta, teacher_assign=True)                     # option 1: the single best candidate
    assert int((r1.gt_inds > 0).sum()) == 1
    r13 = a.assign(bp2, pr2, gt, gl, meta)
    assert int((r13.gt_inds > 0).sum()) == 13
    best = int(torch.argmax(r13.assign_metrics))
    assert int(r1.gt_inds[best]) == 1
    with pytest.raises(RuntimeError, match="2048 queries"):
        a.assign(torch.rand(2049, 4).cuda(), torch.rand(2049, 2).cuda(), gt, gl, meta)
