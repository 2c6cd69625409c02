#!/usr/bin/env python
"""Run bench.module_bench (one encoder MSDeformAttn layer fwd+bwd, fused vs op-by-op) -- for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import semi_detr_amd  # noqa: F401
print(bench.module_bench(torch.device("cuda:0"), iters=int(sys.argv[1]) if len(sys.argv) > 1 else 5))
