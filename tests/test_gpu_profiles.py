"""GPU: the committed traffic figures (profiles/r04_pmc_traffic.json, written by tools/measure_traffic.py) must belong to
the kernels the library launches TODAY for those shapes -- a renamed or re-dispatched kernel would otherwise leave a stale
`roofline.traffic` in the bench line (bench.py drops an entry whose kernel list differs, this test makes the staleness
visible in the suite)."""
import json
import os
import re

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LEVELS = [(100, 167), (50, 84), (25, 42), (13, 21)]


def test_traffic_json_matches_the_kernels_the_library_launches():
    import semi_detr_amd as sda
    import MultiScaleDeformableAttention as MSDA
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")))
    lib = sda._lib.lib()
    dev = torch.device("cuda:0")
    shapes = torch.as_tensor(LEVELS, dtype=torch.long, device=dev)
    starts = torch.cat([shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]])
    S, M, D, L, P = int((shapes[:, 0] * shapes[:, 1]).sum()), 8, 32, 4, 4
    checked = 0
    for group, entry in pmc.items():
        m = re.fullmatch(r"msda_(fwd|bwd)_(enc|dec|micro)_bs(\d+)_Lq(\d+)(_cold|_window)?", group)
        if not m:
            continue
        # the encoder forward has two kernels (include/semidetr_hip.h: semidetr_msda_set_forward_policy): one entry each
        sda._lib.set_forward_policy("window" if m.group(5) == "_window" else "patch")
        direction, n, lq = m.group(1), int(m.group(3)), int(m.group(4))
        value = torch.rand(n, S, M, D, device=dev)
        loc = torch.rand(n, lq, M, L, P, 2, device=dev)
        attn = torch.rand(n, lq, M, L, P, device=dev)
        if direction == "fwd":
            MSDA.ms_deform_attn_forward(value, shapes, starts, loc, attn, 64)
        else:
            MSDA.ms_deform_attn_backward(value, shapes, starts, loc, attn, torch.rand(n, lq, M * D, device=dev), 64)
        got = lib.semidetr_msda_last_kernels().decode().split("+")
        assert got == entry["kernels"], (group, got, entry["kernels"])
        assert entry["hbm_bytes_corrected"] > 0
        checked += 1
    torch.cuda.synchronize()
    sda._lib.set_forward_policy("adaptive")
    assert checked >= 6
