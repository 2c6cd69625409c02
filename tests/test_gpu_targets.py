"""GPU: the batched `get_targets` drop-in against a per-image loop that follows the
reference's `_get_target_single` line by line (detr_od/models/dense_heads/dino_detr_ssod_head.py:1069-1205) on top
of the single-problem `assign()` paths, which are themselves pinned to the reference's fixtures
(tests/test_gpu_matcher.py, tests/test_gpu_o2m.py).  7 decoder layers x 5 images, both branches (Hungarian after
warm-up, one-to-many during warm-up), per-layer `get_targets` and all-layer `get_targets_layers`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NL, B, Q, C = 7, 5, 900, 80


def _inputs(seed):
    g = torch.Generator().manual_seed(seed)
    rng = np.random.default_rng(seed)
    cls = (torch.randn(NL, B, Q, C, generator=g) * 2 - 2).cuda()
    box = torch.cat([torch.rand(NL, B, Q, 2, generator=g), torch.rand(NL, B, Q, 2, generator=g) * 0.4 + 0.02], -1).cuda()
    gts, labs, metas = [], [], []
    shapes = [(800, 1333, 3), (800, 1201, 3), (750, 1333, 3), (800, 1066, 3), (704, 1333, 3)]
    for b in range(B):
        G = [7, 1, 15, 0, 30][b] if seed % 2 == 0 else int(rng.integers(1, 16))
        h, w, _ = shapes[b]
        xy = torch.rand(G, 2, generator=g) * torch.tensor([w * 0.7, h * 0.7])
        wh = torch.rand(G, 2, generator=g) * torch.tensor([w * 0.25, h * 0.25]) + 16
        gts.append(torch.cat([xy, xy + wh], -1).cuda())
        labs.append(torch.randint(0, C, (G,), generator=g).cuda())
        metas.append(dict(img_shape=shapes[b]))
    return cls, box, gts, labs, metas


def _xyxy_to_cxcywh(b):
    return torch.stack([(b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1]], -1)


def _single_hungarian(asg, cls, box, gt, lab, meta):
    """dino_detr_ssod_head.py:1168-1205 with PseudoSampler (pseudo_sampler.py:35-41)."""
    res = asg.assign(box, cls, gt, lab, meta)
    pos = torch.nonzero(res.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
    neg = torch.nonzero(res.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
    labels = gt.new_full((Q,), C, dtype=torch.long)
    labels[pos] = lab[res.gt_inds[pos] - 1].long()
    lw = gt.new_ones(Q)
    bt, bw = torch.zeros_like(box), torch.zeros_like(box)
    bw[pos] = 1.0
    h, w, _ = meta["img_shape"]
    factor = box.new_tensor([w, h, w, h]).unsqueeze(0)
    bt[pos] = _xyxy_to_cxcywh(gt[res.gt_inds[pos] - 1] / factor)
    return labels, lw, bt, bw, pos, neg


def _single_warmup(asg, cls, box, gt, lab, meta):
    """dino_detr_ssod_head.py:1108-1165."""
    res = asg.assign(box, cls.sigmoid(), gt, lab, meta)
    ious = res.max_overlaps.clone()
    ious[ious == -100000000] = 0
    met = res.assign_metrics
    pos = torch.nonzero(res.gt_inds > 0, as_tuple=False).squeeze(-1).unique()
    neg = torch.nonzero(res.gt_inds == 0, as_tuple=False).squeeze(-1).unique()
    pag = res.gt_inds[pos] - 1
    labels = gt.new_full((Q,), C, dtype=torch.long)
    lw = gt.new_ones(Q)
    bt, bw = torch.zeros_like(box), torch.zeros_like(box)
    h, w, _ = meta["img_shape"]
    factor = box.new_tensor([w, h, w, h]).unsqueeze(0)
    bt[pos, :] = _xyxy_to_cxcywh(gt[pag] / factor)
    labels[pos] = lab[pag].long()
    nm = met.new_zeros(Q)
    for gi in torch.unique(pag):
        idx = pos[pag == gi]
        nm[idx] = met[idx] / (met[idx].max() + 10e-8) * ious[idx].max()
    bw[pos, :] = nm[pos].unsqueeze(-1)
    return labels, lw, bt, bw, nm, pos, neg


@pytest.mark.parametrize("seed", [0, 1])
@pytest.mark.parametrize("warm_up", [False, True])
def test_get_targets_equals_per_image_loop(seed, warm_up):
    from semi_detr_amd import TargetAssigner
    from semi_detr_amd import targets as tg
    head = TargetAssigner(num_classes=C, in_warm_up=warm_up)
    cls, box, gts, labs, metas = _inputs(seed)
    single = _single_warmup if warm_up else _single_hungarian
    asg = head.assigner1 if warm_up else head.assigner2
    layers = head.get_targets_layers(cls, box, gts, labs, metas)
    assert len(layers) == NL
    for li in range(NL):
        want = [single(asg, cls[li, b], box[li, b], gts[b], labs[b], metas[b]) for b in range(B)]
        per_layer = head.get_targets([cls[li, b] for b in range(B)], [box[li, b] for b in range(B)], gts, labs,
                                     None, metas)
        for got in (layers[li], per_layer):
            assert len(got) == (7 if warm_up else 6)
            n_lists = 5 if warm_up else 4
            for k in range(n_lists):
                assert isinstance(got[k], list) and len(got[k]) == B
                for b in range(B):
                    w = want[b][k]
                    if w.dtype == torch.long:
                        assert torch.equal(got[k][b], w), (li, b, k)
                    else:
                        assert torch.allclose(got[k][b], w, rtol=0, atol=2e-6), (li, b, k)
            num_pos = sum(int(w[-2].numel()) for w in want)
            num_neg = sum(int(w[-1].numel()) for w in want)
            assert got[-2] == num_pos and got[-1] == num_neg and isinstance(got[-2], int)
    tg.check_deferred(block=True)          # nothing invalid was seen


def test_get_targets_reports_invalid_costs_deferred():
    """A NaN logit makes scipy raise ValueError at hungarian_assigner.py:136; here the status word is inspected at a
    later call (or on demand) instead of draining the stream inside get_targets."""
    from semi_detr_amd import TargetAssigner
    from semi_detr_amd import targets as tg
    head = TargetAssigner(num_classes=C)
    cls, box, gts, labs, metas = _inputs(3)
    bad = cls[0].clone()
    bad[1, 5, :] = float("nan")
    tg.check_deferred(block=True)
    head.get_targets(list(bad), list(box[0]), gts, labs, None, metas)            # does not raise yet
    with pytest.raises(ValueError, match="invalid numeric entries"):
        tg.check_deferred(block=True)
    with pytest.raises(ValueError, match="invalid numeric entries"):
        head.get_targets(list(bad), list(box[0]), gts, labs, None, metas, check=True)
    # an out-of-range class label is an index error in the reference; here it poisons the cost (NaN) -> same report
    labs2 = [l.clone() for l in labs]
    labs2[0][0] = C + 3
    with pytest.raises(ValueError, match="invalid numeric entries"):
        head.get_targets(list(cls[0]), list(box[0]), gts, labs2, None, metas, check=True)
    tg.check_deferred(block=True)
