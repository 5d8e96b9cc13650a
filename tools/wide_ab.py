#!/usr/bin/env python3
"""Round 4: gemm_wide_kernel reading the decode copy with raw x staged by LDS DMA (default plan of a plain layer that carries its copy: "wide_copy")
against the same kernel on the checkpoint rows with register-staged, slot-ordered x (tuning.reserved[3] = 46) and the 128 x 256 kernel (44); interleaved
rounds on rotating layers in a hipGraph.  Usage: python tools/wide_ab.py [--ms 4096] [--shapes 4096x4096,...] [--dtype f16]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="4096")
ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float16 if a.dtype == "f16" else torch.bfloat16


def tune(v):
    if v is None:
        return None
    t = _lib.GptqTuning()
    t.path, t.reserved[3] = 3, v
    return t


for shp in a.shapes.split(","):
    K, N = map(int, shp.split("x"))
    ls = [make_layer(K, N, dev, dtype=dt, seed=i) for i in range(6)]
    for M in map(int, a.ms.split(",")):
        x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
        best = {}
        for _ in range(a.rounds):
            for name, v in (("copy + DMA x (default)", None), ("checkpoint rows", 46), ("128 x 256 tiles", 44)):
                s = run(ls, x, tune(v), reps=3)
                best[name] = min(best.get(name, 1e9), s)
        plan = _lib.describe_plan(ls[0]._layer, M)
        print(f"{K}x{N} M={M} {a.dtype} [{plan.get('kernel')}]: " + "   ".join(f"{k}: {v * 1e6:7.1f} us {2 * M * K * N / v / 1e12:6.0f} TF" for k, v in best.items()), flush=True)
    del ls
