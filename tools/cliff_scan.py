#!/usr/bin/env python3
"""Default-plan timing over a grid (bits x dtype x act-order x group size) x M x shape: one line per configuration, us per layer call for
every M -- a row whose numbers jump against its neighbours is a planner / kernel cliff.  Usage: python tools/cliff_scan.py [--slice A|B|C|D]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--slice", default="A")
ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
a = ap.parse_args()
dev = torch.device("cuda:0")
F16, BF16, F32 = torch.float16, torch.bfloat16, torch.float32
SL = {"A": ([(4, BF16, False, 128), (4, BF16, True, 128)], (1, 2, 3, 4, 5, 8, 16, 32, 64)),
      "B": ([(4, F16, False, 32), (4, F16, False, 64), (4, F16, True, 32)], (1, 2, 3, 4, 5, 8, 16, 32, 64)),
      "C": ([(3, F16, False, 32), (3, F16, True, 32), (8, F16, False, 32), (8, F16, True, 32), (2, F16, False, 64)], (1, 2, 4, 8, 16, 64)),
      "D": ([(4, F32, False, 128), (4, F32, True, 128), (8, F32, False, 32)], (1, 2, 4, 8, 16, 64)),
      "F": ([(4, F16, False, 128), (4, F16, True, 128)], (1, 2, 3, 4, 5, 8, 16, 32, 64)),
      "E": ([(4, F16, False, 128), (4, F16, True, 128)], (65, 96, 128, 192, 256, 384, 512, 1024))}
cfgs, Ms = SL[a.slice]
for shp in a.shapes.split(","):
    K, N = map(int, shp.split("x"))
    for bits, dt, act, gs in cfgs:
        nl = max(3, min(24, (320 << 20) // (K * N * bits // 8)))
        ls = [make_layer(K, N, dev, bits=bits, gs=gs, act_order=act, dtype=dt, seed=i) for i in range(nl)]
        out = []
        for M in Ms:
            x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
            d = _lib.describe_plan(ls[0]._layer, M)
            out.append(f"{M}:{run(ls, x, None, reps=3) * 1e6:.1f}[{str(d.get('kernel'))[:6]}]")
        print(f"{K}x{N} b{bits} {str(dt)[6:]:8s} act={int(act)} g{gs}:  " + "  ".join(out), flush=True)
        del ls
