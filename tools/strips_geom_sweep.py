#!/usr/bin/env python3
"""Decode-copy kernel geometries on shapes outside the Llama-7B set: default plan against forced (waves, chunks in flight) and two strips per workgroup,
HBM-cold rotating layers in a hipGraph (bench.py's protocol).  usage: python tools/strips_geom_sweep.py [--shapes 5120x5120,...] [--ms 1,2,4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from autogptq_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="5120x5120,13824x5120,5120x13824,8192x8192,6656x6656,17920x6656,6656x17920")
ap.add_argument("--ms", default="1,2,4")
ap.add_argument("--act", action="store_true")
ap.add_argument("--bits", type=int, default=4)
ap.add_argument("--gs", type=int, default=128)
args = ap.parse_args()
dev = torch.device("cuda:0")


def tune(waves, u, nstr=0, ks=0):
    t = _lib.GptqTuning()
    t.path = 8
    t.waves, t.ksplit = waves, ks
    t.reserved[_lib.LAB.DEPTH] = u
    t.reserved[1] = nstr
    return t


def time_layers(ls, x, t, reps=6):
    def call():
        return [q(x, tuning=t) if t is not None else q(x) for _, _, _, q in ls]
    try:
        with torch.no_grad():
            call()
    except Exception as e:
        return None, str(e)[:60]
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        outs = call()
    bench.settle(g, dev)
    _, evt = bench.time_graph(g, reps, dev)
    del g, outs
    return evt / (reps * len(ls)) * 1e6, ""


for shp in args.shapes.split(","):
    K, N = (int(v) for v in shp.split("x"))
    n = max(4, -(-(320 << 20) // (K * N * args.bits // 8)))
    ls = [("b", K, N, bench.make_layer(K, N, dev, bits=args.bits, gs=args.gs, act_order=args.act, seed=9300 + i)) for i in range(n)]
    for M in (int(m) for m in args.ms.split(",")):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        base, _ = time_layers(ls, x, None)
        plan = bench._plan_dict(ls, K, N, M)
        row = [f"default[{plan.get('kernel')} w={plan.get('waves')} u={plan.get('u')} ks={plan.get('ksplit')}] {base:6.2f}"]
        for name, t in (("16x2", tune(16, 2)), ("8x4", tune(8, 4)), ("4x4", tune(4, 4)), ("8x2", tune(8, 2)), ("2str8x4", tune(8, 4, 2)), ("2str4x4", tune(4, 4, 2)), ("1str4x4", tune(4, 4, 1)), ("1str8x2", tune(8, 2, 1)), ("1str4x2", tune(4, 2, 1)), ("1str2x4", tune(2, 4, 1))):
            us, err = time_layers(ls, x, t)
            row.append(f"{name} {us:6.2f}" if us is not None else f"{name} refused")
        print(f"{K}x{N} M={M}: " + " | ".join(row), flush=True)
    del ls
    torch.cuda.empty_cache()
