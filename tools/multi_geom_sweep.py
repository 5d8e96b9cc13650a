#!/usr/bin/env python3
"""gptq_forward_multi (q|k|v, gate|up: layers that share x in ONE decode launch) across model families: default plan against forced geometries of the
decode-copy kernel, HBM-cold rotating groups in a hipGraph.  usage: python tools/multi_geom_sweep.py [--ms 1,2,4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from autogptq_amd import _lib  # noqa: E402
from autogptq_amd.qlinear_mi355x import forward_multi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="1,2,4")
args = ap.parse_args()
dev = torch.device("cuda:0")
GROUPS = [("7B qkv", 4096, (4096, 4096, 4096)), ("7B gate|up", 4096, (11008, 11008)), ("13B qkv", 5120, (5120, 5120, 5120)), ("13B gate|up", 5120, (13824, 13824)),
          ("30B qkv", 6656, (6656, 6656, 6656)), ("30B gate|up", 6656, (17920, 17920)), ("70B qkv (GQA)", 8192, (8192, 1024, 1024)), ("70B gate|up", 8192, (28672, 28672)),
          ("70B TP8 qkv", 8192, (1024, 128, 128)), ("70B TP8 gate|up", 8192, (3584, 3584)), ("8B qkv (GQA)", 4096, (4096, 1024, 1024)), ("8B gate|up", 4096, (14336, 14336))]


def tune(waves, u, nstr=0):
    t = _lib.GptqTuning()
    t.path = 8
    t.waves = waves
    t.reserved[_lib.LAB.DEPTH] = u
    t.reserved[1] = nstr
    return t


def time_groups(gs, x, t, reps=6):
    def call():
        return [forward_multi(g, x, t) for g in gs]
    try:
        with torch.no_grad():
            call()
    except Exception as e:
        return None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        outs = call()
    bench.settle(g, dev)
    _, evt = bench.time_graph(g, reps, dev)
    del g, outs
    return evt / (reps * len(gs)) * 1e6


for name, K, widths in GROUPS:
    nbytes = sum(K * n // 2 for n in widths)
    n = max(3, min(24, -(-(320 << 20) // nbytes)))
    gs = [[bench.make_layer(K, w, dev, seed=9500 + 8 * i + j) for j, w in enumerate(widths)] for i in range(n)]
    for M in (int(m) for m in args.ms.split(",")):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        base = time_groups(gs, x, None)
        row = [f"default {base:6.2f}"]
        for nm, t in (("16x2", tune(16, 2, 1)), ("8x2", tune(8, 2, 1)), ("8x4", tune(8, 4, 1)), ("4x4", tune(4, 4, 1)), ("2str8x4", tune(8, 4, 2)), ("2str4x4", tune(4, 4, 2))):
            us = time_groups(gs, x, t)
            row.append(f"{nm} {us:6.2f}" if us is not None else f"{nm} refused")
        print(f"{name:18s} K={K} N={'+'.join(str(w) for w in widths)} ({sum(widths) // 16} strips) M={M}: " + " | ".join(row), flush=True)
    del gs
    torch.cuda.empty_cache()
