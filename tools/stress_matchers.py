#!/usr/bin/env python
"""Randomised sweeps of the index-producing kernels against their CPU counterparts (not part of the test suite):
class-aware NMS and the one-to-many assigner against the C oracle, the Hungarian solver against scipy.
Everything must agree EXACTLY in indices / labels / order.
    python tools/stress_matchers.py --cases 150 --seed 0"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from scipy.optimize import linear_sum_assignment as scipy_lsa  # noqa: E402

import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    import semi_detr_amd as sda
    rng = np.random.default_rng(a.seed)
    bad = {"nms": 0, "o2m": 0, "lsap": 0}
    for case in range(a.cases):
        # ---- NMS
        B, Q, C = int(rng.integers(1, 4)), int(rng.integers(1, 1500 if case % 4 == 0 else 700)), int(rng.integers(1, 60))
        logits = rng.normal(rng.uniform(-6, 1), rng.uniform(0.5, 3), (B, Q, C)).astype(np.float32)
        if rng.random() < 0.3:
            logits = (np.round(logits * 8) / 8).astype(np.float32)
        k = max(Q // 6, 1)
        cxcy, wh = rng.random((B, Q, 2)), rng.random((B, Q, 2)) * 0.4 + 0.01
        for b in range(B):
            src = rng.integers(0, k, Q - k)
            cxcy[b, k:] = cxcy[b, src] + rng.normal(0, 0.02, (Q - k, 2))
            wh[b, k:] = wh[b, src] * (1 + rng.normal(0, 0.08, (Q - k, 2)))
        bbox = np.concatenate([cxcy, wh], -1).astype(np.float32)
        shapes = [(int(rng.integers(200, 1000)), int(rng.integers(200, 1400))) for _ in range(B)]
        mp = int(rng.choice([1, 10, 100, 300, 2048]))
        thr, iou = float(rng.choice([0.01, 0.05, 0.3])), float(rng.choice([0.3, 0.6, 0.9]))
        res = sda.get_bboxes_for_pseudo_label(torch.from_numpy(logits).cuda(), torch.from_numpy(bbox).cuda(),
                                              [dict(img_shape=(h, w, 3)) for h, w in shapes], score_thr=thr,
                                              iou_threshold=iou, max_per_img=mp)
        for b in range(B):
            ed, el = oracle.pseudo_nms(logits[b], bbox[b], shapes[b][0], shapes[b][1], score_thr=thr, iou_thr=iou, max_num=mp)
            d, lab = res[b][0].cpu().numpy(), res[b][1].cpu().numpy()
            if d.shape != ed.shape or not np.array_equal(lab, el) or not np.array_equal(d[:, :4], ed[:, :4]):
                bad["nms"] += 1
                print("NMS mismatch", case, b, Q, C, mp, thr, iou, d.shape, ed.shape)
        # ---- O2M
        Q, C, G = int(rng.integers(13, 1200)), int(rng.integers(1, 90)), int(rng.integers(0, 60))
        iw, ih = int(rng.integers(200, 1400)), int(rng.integers(200, 1000))
        gt = np.zeros((G, 4), np.float32)
        gt[:, :2] = rng.random((G, 2)) * [iw * 0.7, ih * 0.7]
        gt[:, 2:] = gt[:, :2] + rng.random((G, 2)) * [iw * 0.3, ih * 0.3] + 4
        bp = np.concatenate([rng.random((Q, 2)), rng.random((Q, 2)) * 0.4 + 0.02], -1).astype(np.float32)
        if G:
            src = rng.integers(0, G, Q // 2)
            n = gt[src] / np.asarray([iw, ih, iw, ih], np.float32)
            near = np.stack([(n[:, 0] + n[:, 2]) / 2, (n[:, 1] + n[:, 3]) / 2, n[:, 2] - n[:, 0], n[:, 3] - n[:, 1]], -1)
            bp[:Q // 2] = (near * (1 + rng.normal(0, 0.1, near.shape))).clip(0.001, 0.999)
        prob = (rng.random((Q, C)) ** 2).astype(np.float32)
        gl = rng.integers(0, C, G)
        topk = int(rng.choice([1, 5, 13]))
        out = sda.O2MAssigner(candidate_topk=topk).assign_batch(
            torch.from_numpy(bp).cuda()[None], torch.from_numpy(prob).cuda()[None], [torch.from_numpy(gt).cuda()],
            [torch.from_numpy(gl).cuda()], [dict(img_shape=(ih, iw, 3))])
        gi, lab, mo, am = oracle.o2m_assign(bp, prob, gt, gl, iw, ih, topk=topk)
        if not (np.array_equal(out["gt_inds"][0].cpu().numpy(), gi) and np.array_equal(out["labels"][0].cpu().numpy(), lab)
                and np.array_equal(out["max_overlaps"][0].cpu().numpy(), mo)):
            bad["o2m"] += 1
            print("O2M mismatch", case, Q, C, G, topk)
        # ---- LSAP
        Q, G = int(rng.integers(1, 400)), int(rng.integers(1, 120))
        cost = rng.normal(0, 1, (Q, G)).astype(np.float32)
        if rng.random() < 0.4:
            cost = np.round(cost * 2) / 2                    # heavy ties
        if rng.random() < 0.2:
            cost[rng.random((Q, G)) < 0.1] = np.inf
        try:
            r, c = scipy_lsa(cost.astype(np.float64))
            rr, cc = sda.linear_sum_assignment(torch.from_numpy(cost).cuda())
            if not (np.array_equal(rr.cpu().numpy(), r) and np.array_equal(cc.cpu().numpy(), c)):
                bad["lsap"] += 1
                print("LSAP mismatch", case, Q, G)
        except ValueError as e:
            try:
                sda.linear_sum_assignment(torch.from_numpy(cost).cuda())
                bad["lsap"] += 1
                print("LSAP: scipy raised, kernel did not", case, e)
            except ValueError:
                pass
    print("cases", a.cases, "mismatches", bad)


if __name__ == "__main__":
    main()
