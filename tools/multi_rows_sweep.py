#!/usr/bin/env python3
"""gptq_forward_multi at batched-decode / short-prompt row counts against the same layers called one by one (each on its own best plan), across model families;
HBM-cold rotating groups in a hipGraph.  A group call that loses to its separate calls is a planner rule to fix.  usage: python tools/multi_rows_sweep.py [--ms 8,16,32,64,128,256]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from autogptq_amd import _lib  # noqa: E402
from autogptq_amd.qlinear_mi355x import forward_multi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="5,8,16,32,64,128,256")
args = ap.parse_args()
dev = torch.device("cuda:0")
GROUPS = [("7B qkv", 4096, (4096, 4096, 4096)), ("7B gate|up", 4096, (11008, 11008)), ("13B qkv", 5120, (5120, 5120, 5120)), ("13B gate|up", 5120, (13824, 13824)),
          ("70B qkv (GQA)", 8192, (8192, 1024, 1024)), ("70B TP8 qkv", 8192, (1024, 128, 128)), ("70B TP8 gate|up", 8192, (3584, 3584)), ("8B qkv (GQA)", 4096, (4096, 1024, 1024)),
          ("8B gate|up", 4096, (14336, 14336))]


def timed(fn, n_groups, reps=6):
    with torch.no_grad():
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        outs = fn()
    bench.settle(g, dev)
    _, evt = bench.time_graph(g, reps, dev)
    del g, outs
    return evt / (reps * n_groups) * 1e6


for name, K, widths in GROUPS:
    nbytes = sum(K * n // 2 for n in widths)
    n = max(3, min(24, -(-(320 << 20) // nbytes)))
    gs = [[bench.make_layer(K, w, dev, seed=9700 + 8 * i + j) for j, w in enumerate(widths)] for i in range(n)]
    for M in (int(m) for m in args.ms.split(",")):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        t_multi = timed(lambda: [forward_multi(g, x) for g in gs], n)
        t_sep = timed(lambda: [[l(x) for l in g] for g in gs], n)
        plans = "+".join(_lib.describe_plan(l._layer, M)["kernel"] for l in gs[0])
        flag = "   <-- the group call LOSES" if t_multi > 1.03 * t_sep else ""
        print(f"{name:16s} K={K} N={'+'.join(str(w) for w in widths)} M={M:4d}: forward_multi {t_multi:7.2f} us | one by one {t_sep:7.2f} us [{plans}]{flag}", flush=True)
    del gs
    torch.cuda.empty_cache()
