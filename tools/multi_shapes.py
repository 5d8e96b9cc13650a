#!/usr/bin/env python3
"""gptq_forward_multi groups of larger models (13B q|k|v and gate|up, 70B GQA q|k|v and gate|up) against separate launches, M = 1 / 4 / 16.
Usage: python tools/multi_shapes.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.stream_sweep import timed
from autogptq_amd.qlinear_mi355x import forward_multi

dev = torch.device("cuda:0")
GROUPS = (("7B qkv", 4096, (4096,) * 3), ("7B gate|up", 4096, (11008,) * 2), ("13B qkv", 5120, (5120,) * 3), ("13B gate|up", 5120, (13824,) * 2),
          ("70B qkv (GQA)", 8192, (8192, 1024, 1024)), ("70B gate|up", 8192, (28672,) * 2), ("70B TP8 gate|up shard", 8192, (3584,) * 2))
for name, K, Ns in GROUPS:
    ng = max(2, min(12, (400 << 20) // (K * sum(Ns) // 2)))
    groups = [[make_layer(K, n, dev, seed=100 * gi + i) for i, n in enumerate(Ns)] for gi in range(ng)]
    out = []
    for M in (1, 4, 16):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        a, _ = timed(lambda: [forward_multi(g, x) for g in groups])
        b, _ = timed(lambda: [[q(x) for q in g] for g in groups])
        out.append(f"M={M}: multi {a / ng * 1e6:6.2f} us, separate {b / ng * 1e6:6.2f} us")
    print(f"{name:24s} " + "   ".join(out), flush=True)
    del groups
