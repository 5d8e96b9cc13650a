#!/usr/bin/env python3
"""Round 5: the stream-K form of the wide prefill tile (csrc/gemm_wide_sk.hip, tuning.reserved[3] = 48) against the planner's choice without it (49: 128 x 256
tiles, two K groups, balanced tail, mid kernel, wide_copy ... whatever plan_gemm picks), layer call = x permute (act-order) + GEMM, interleaved rounds on
rotating layers in a hipGraph.  Usage: python tools/wide_sk_ab.py [--ms 2048,...] [--shapes 4096x4096,...] [--dtype f16] [--act 0,1]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="128,256,512,1024,1536,2048,2304,3072,4096,8192")
ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--act", default="0,1")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--layers", type=int, default=6)
ap.add_argument("--bits", type=int, default=4)
ap.add_argument("--gs", type=int, default=128)
ap.add_argument("--dv", default="0", help="lab: DMA placement variants of the stream-K kernel to time (tuning.reserved[0])")
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float16 if a.dtype == "f16" else torch.bfloat16


def tune(v, dv=0):
    t = _lib.GptqTuning()
    t.path, t.reserved[3], t.reserved[0] = 3, v, dv
    return t


# settle the clocks first (bench.py does the same): ~0.3 s of MFMA work
warm = [make_layer(4096, 4096, dev, dtype=dt, seed=99)]
xw = (torch.rand(4096, 4096, device=dev) - 0.5).to(dt)
for _ in range(3):
    run(warm, xw, None, reps=200)
del warm, xw

for shp in a.shapes.split(","):
    K, N = map(int, shp.split("x"))
    for act in map(int, a.act.split(",")):
        ls = [make_layer(K, N, dev, bits=a.bits, gs=a.gs, dtype=dt, seed=i, act_order=bool(act)) for i in range(a.layers)]
        for M in map(int, a.ms.split(",")):
            x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
            best = {}
            names = {}
            for _ in range(a.rounds):
                for name, v, dv in [("without", 49, 0)] + [("stream-K" + (f" dv{d}" if d else ""), 48, int(d)) for d in a.dv.split(",")]:
                    t = tune(v, dv)
                    names[name] = _lib.describe_plan(ls[0]._layer, M, t).get("kernel")
                    s = run(ls, x, t, reps=3)
                    best[name] = min(best.get(name, 1e9), s)
            d = _lib.describe_plan(ls[0]._layer, M)
            w = best.pop("without")
            print(f"int{a.bits} g{a.gs} {K}x{N} M={M:5d} {a.dtype} act={act} default={d.get('kernel'):9s} | without [{names['without']:9s}] {w * 1e6:8.1f} us {2 * M * K * N / w / 1e12:6.0f} TF | " +
                  " | ".join(f"{k} {s * 1e6:8.1f} us {2 * M * K * N / s / 1e12:6.0f} TF {w / s:5.2f}x" for k, s in best.items()), flush=True)
            del x
        del ls
