#!/usr/bin/env python3
"""17 ... 256 rows on the deep / largest layers (where the rows kernel's 64 Mi-weight limit and the panel kernel's depth limits leave the round-2/3 kernels in
charge): default plan against every kernel family forced through the lab knobs, HBM-cold rotating layers.  usage: python tools/mid_band_sweep.py [--shapes ...] [--ms ...]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from autogptq_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="28672x8192,17920x6656,13824x5120,14336x4096,11008x4096,8192x28672,6656x17920,5120x13824")
ap.add_argument("--ms", default="24,32,48,64,96,128,192,256")
ap.add_argument("--act", action="store_true")
ap.add_argument("--bits", type=int, default=4)
ap.add_argument("--gs", type=int, default=128)
args = ap.parse_args()
dev = torch.device("cuda:0")
L = _lib.LAB


def tk(kernel=0, variant=0, g0=0):
    t = _lib.GptqTuning()
    t.path = 3
    t.reserved[L.GEMM_KERNEL] = kernel
    t.reserved[L.GEMM_VARIANT] = variant
    t.reserved[0] = g0
    return t


def timeit(ls, x, t):
    def call():
        return [q(x, tuning=t) if t is not None else q(x) for _, _, _, q in ls]
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
    _, evt = bench.time_graph(g, 5, dev)
    del g, outs
    return evt / (5 * len(ls)) * 1e6


for shp in args.shapes.split(","):
    K, N = (int(v) for v in shp.split("x"))
    n = max(4, -(-(320 << 20) // (K * N * args.bits // 8)))
    ls = [("b", K, N, bench.make_layer(K, N, dev, bits=args.bits, gs=args.gs, act_order=args.act, seed=9950 + i)) for i in range(n)]
    for M in (int(m) for m in args.ms.split(",")):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        timeit(ls, x, None)
        row = []
        for nm, t in (("default", None), ("tiled", tk(L.GEMM_TILED)), ("mid", tk(L.GEMM_MID)), ("stream64", tk(L.GEMM_STREAM64)), ("rows", tk(0, L.VARIANT_ROWS_ON)),
                      ("panel", tk(0, L.VARIANT_PANEL_ON)), ("wide_sk", tk(0, L.VARIANT_WIDE_SK_ON))):
            us = timeit(ls, x, t)
            kern = _lib.describe_plan(ls[0][3]._layer, M, t).get("kernel") if us is not None else "-"
            row.append(f"{nm}[{kern}] {us:6.1f}" if us is not None else f"{nm} refused")
        print(f"{K}x{N} M={M:3d}: " + " | ".join(row), flush=True)
    del ls
    torch.cuda.empty_cache()
