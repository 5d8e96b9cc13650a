#!/usr/bin/env python3
"""[gate | up] layers with the SiLU * mul epilogue (the pair form of the decode-copy kernel: strip s of both halves per workgroup) across families: default against
forced (waves, chunks in flight); HBM-cold rotating layers.  usage: python tools/pair_geom_sweep.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import autogptq_amd  # noqa: E402
import bench  # noqa: E402
from autogptq_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")


def mk(K, N, sd):
    g = torch.Generator(device=dev).manual_seed(sd)
    q = autogptq_amd.QuantLinear(4, 128, K, N, False, epilogue="silu_mul")
    G = K // 128
    q.qweight = torch.randint(-2**31, 2**31 - 1, (K // 8, N), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    q.qzeros = torch.randint(-2**31, 2**31 - 1, (G, N // 8), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    q.scales = (0.002 * (1 + 0.1 * torch.rand(G, N, device=dev, generator=g))).half()
    q = q.to(dev)
    q.post_init()
    return q


def tune(w, u):
    t = _lib.GptqTuning()
    t.path, t.waves = 8, w
    t.reserved[_lib.LAB.DEPTH] = u
    return t


def timeit(ls, x, t):
    def call():
        return [q(x, tuning=t) if t is not None else q(x) for q in ls]
    try:
        with torch.no_grad():
            call()
    except Exception:
        return None
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        outs = call()
    bench.settle(g, dev)
    _, evt = bench.time_graph(g, 6, dev)
    del g, outs
    return evt / (6 * len(ls)) * 1e6


for name, K, I in (("7B", 4096, 11008), ("13B", 5120, 13824), ("8B", 4096, 14336), ("30B", 6656, 17920), ("70B TP8", 8192, 3584), ("70B", 8192, 28672)):
    n = max(3, min(16, -(-(320 << 20) // (K * I))))
    ls = [mk(K, 2 * I, 9400 + i) for i in range(n)]
    for M in (1, 2, 4):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        timeit(ls, x, None)
        p = _lib.describe_plan(ls[0]._layer, M)
        row = [f"default[{p.get('kernel')} w={p.get('waves')} u={p.get('u')} pair={p.get('pair')}] {timeit(ls, x, None):6.2f}"]
        for w, u in ((16, 2), (8, 2), (8, 4), (4, 4), (4, 2), (16, 4)):
            us = timeit(ls, x, tune(w, u))
            row.append(f"{w}x{u} {us:6.2f}" if us else f"{w}x{u} -")
        print(f"{name:8s} {K}->2x{I} M={M}: " + " | ".join(row), flush=True)
    del ls
    torch.cuda.empty_cache()
