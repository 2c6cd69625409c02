#!/usr/bin/env python
"""Randomised parity sweep (not part of the test suite): many small random MSDA configurations -- pyramids, heads,
points, query counts, encoder (Lq == S) and decoder style, in-range / out-of-range samples -- HIP kernels against
the CPU oracle.  Prints the worst errors and any configuration beyond tolerance.
    python tools/stress_parity.py --cases 300 --seed 0"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import os as _os
_os.environ.setdefault("SEMIDETR_EXPERIMENTS", "1")
import torch  # noqa: E402

import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--first", type=int, default=0, help="skip (but generate) the cases before this one")
    ap.add_argument("--only", type=int, default=-1, help="run just this case of the sequence (the others only advance the generator)")
    ap.add_argument("--verbose", action="store_true", help="print every case before it runs (to find a crashing one)")
    ap.add_argument("--variant", type=int, nargs=2, default=[0, 0], help="forced (fwd, bwd) kernel variants; cases a forced "
                    "variant does not apply to are skipped")
    ap.add_argument("--policy", default="adaptive", choices=["adaptive", "patch", "window"],
                    help="encoder forward kernel (semidetr_msda_set_forward_policy); \"window\" sends every eligible case through msda_rw_d32")
    a = ap.parse_args()
    import semi_detr_amd  # noqa: F401  (installs the module below)
    semi_detr_amd._lib.set_forward_policy(a.policy)
    import MultiScaleDeformableAttention as MSDA
    semi_detr_amd._lib.set_variant(*a.variant)
    rng = np.random.default_rng(a.seed)
    skipped = 0
    worst = dict(out=0.0, gv=0.0, gl=0.0, ga=0.0)
    bad = []
    for case in range(a.cases):
        L = int(rng.integers(1, 6))
        shapes = [(int(rng.integers(1, 40)), int(rng.integers(1, 50))) for _ in range(L)]
        if rng.random() < 0.5:
            shapes.sort(key=lambda s: -s[0] * s[1])
        M = int(rng.choice([1, 2, 3, 4, 8, 16]))
        P = int(rng.choice([1, 2, 4, 4, 4, 5]))
        N = int(rng.integers(1, 4))
        S = sum(h * w for h, w in shapes)
        enc = rng.random() < 0.5
        Lq = S if enc else int(rng.integers(1, 200) if rng.random() < 0.5 else rng.integers(200, 900))
        shp = np.asarray(shapes, np.int64)
        mode = rng.choice(["in", "wide", "near"])
        if enc and mode == "near":
            ref = np.concatenate([np.stack(np.meshgrid((np.arange(w) + 0.5) / w, (np.arange(h) + 0.5) / h), -1).reshape(-1, 2)
                                  for h, w in shapes])
            loc = ref[None, :, None, None, None, :] + rng.standard_normal((N, S, M, L, P, 2)) * (rng.choice([0.5, 2, 6]) / shp[None, None, None, :, None, ::-1])
        elif mode == "wide":
            loc = rng.random((N, Lq, M, L, P, 2)) * 1.6 - 0.3
        else:
            loc = rng.random((N, Lq, M, L, P, 2))
        value = (rng.random((N, S, M, 32)) - 0.3).astype(np.float32)
        attn = rng.random((N, Lq, M, L, P)) + 1e-5
        attn /= attn.sum((-1, -2), keepdims=True)
        gout = rng.standard_normal((N, Lq, M * 32)).astype(np.float32)
        loc, attn = loc.astype(np.float32), attn.astype(np.float32)
        if (a.only >= 0 and case != a.only) or case < a.first:
            continue
        if a.verbose:
            print("case", case, shapes, "N", N, "M", M, "P", P, "Lq", Lq, "enc", enc, mode, file=sys.stderr, flush=True)
        tv, tl, ta, tg = (torch.from_numpy(x).cuda() for x in (value, loc, attn, gout))
        tsh = torch.from_numpy(shp).cuda()
        tls = torch.cat([tsh.new_zeros(1), (tsh[:, 0] * tsh[:, 1]).cumsum(0)[:-1]])
        try:
            out = MSDA.ms_deform_attn_forward(tv, tsh, tls, tl, ta, 64).cpu().numpy()
            if a.verbose:
                print("   forward done", file=sys.stderr, flush=True)
            gv, gl, ga = (t.cpu().numpy() for t in MSDA.ms_deform_attn_backward(tv, tsh, tls, tl, ta, tg, 64))
        except RuntimeError as e:
            if a.variant != [0, 0] and "need" in str(e):
                skipped += 1
                continue
            raise
        o_out = oracle.msda_forward(value, shp, loc, attn)
        o_gv, o_gl, o_ga = oracle.msda_backward(value, shp, loc, attn, gout)
        errs = dict(out=np.abs(out - o_out).max(), gv=np.abs(gv - o_gv).max() / max(1.0, np.abs(o_gv).max()),
                    gl=np.abs(gl - o_gl).max() / max(1.0, np.abs(o_gl).max()), ga=np.abs(ga - o_ga).max() / max(1.0, np.abs(o_ga).max()))
        for k, v in errs.items():
            worst[k] = max(worst[k], float(v))
        if errs["out"] > 2e-5 or errs["gv"] > 2e-5 or errs["gl"] > 1e-4 or errs["ga"] > 2e-5:
            bad.append((case, shapes, N, M, P, Lq, enc, mode, {k: float(v) for k, v in errs.items()}))
    print("cases", a.cases, "skipped", skipped, "variant", a.variant, "worst", worst)
    for b in bad[:10]:
        print("BEYOND TOLERANCE:", b)
    print("bad", len(bad))


if __name__ == "__main__":
    main()
