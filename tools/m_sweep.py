#!/usr/bin/env python3
"""Default plans across the row counts: us per layer call on HBM-cold rotating layers (bench.py's protocol) for M = 1 ... 1024 on the Llama-7B shapes --
a planner threshold set wrongly shows as a step DOWN at the next row count.  usage: python tools/m_sweep.py [--act] [--bits 4] [--gs 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--act", action="store_true")
ap.add_argument("--bits", type=int, default=4)
ap.add_argument("--gs", type=int, default=128)
ap.add_argument("--ms", default="1,2,3,4,5,6,8,12,16,24,32,48,64,80,96,112,128,160,192,256,320,384,512,640,768,1024")
ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
args = ap.parse_args()
dev = torch.device("cuda:0")
Ms = [int(m) for m in args.ms.split(",")]
for shp in args.shapes.split(","):
    K, N = (int(v) for v in shp.split("x"))
    n = max(4, -(-(320 << 20) // (K * N * args.bits // 8)))
    ls = [("b", K, N, bench.make_layer(K, N, dev, bits=args.bits, gs=args.gs, act_order=args.act, seed=9100 + i)) for i in range(n)]
    prev = None
    print(f"== {K}x{N} int{args.bits} g{args.gs} act={args.act}: {n} rotating layers", flush=True)
    for M in Ms:
        xs = {K: (torch.rand(M, K, device=dev) - 0.5).half()}
        per = bench._time_layers(ls, xs, dev, 6) * 1e6
        plan = bench._plan_dict(ls, K, N, M)
        flag = "   <-- faster than the previous (smaller) row count" if prev is not None and per < prev * 0.97 else ""
        tf = 2.0 * M * K * N / per / 1e6
        print(f"   M={M:5d}: {per:8.2f} us  {tf:7.1f} TF  {plan.get('kernel', '?'):9s} tiles={plan.get('tiles', plan.get('strips', ''))} waves={plan.get('waves')} u={plan.get('u')}{flag}", flush=True)
        prev = per
    del ls
    torch.cuda.empty_cache()
