#!/usr/bin/env python3
"""Round 5: the exchange-free batched-decode kernel (csrc/gemm_rows.hip, tuning.reserved[3] = 50, reserved[0] = RB, reserved[1] = S) against the planner's
choice without it (51), layer call on rotating layers in a hipGraph.  Usage: python tools/rows_ab.py [--ms 16,64,128] [--shapes ...] [--geoms 1x2,2x2,...] [--act 0]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="8,16,32,64,128,256")
ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
ap.add_argument("--geoms", default="0x0,1x1,1x2,1x4,2x1,2x2,2x3,2x4,2x6")
ap.add_argument("--xb", default="2", help="x buffers per wave (2: double-buffered, 8 / 16 waves; 1: single, 16 waves)")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--act", default="0")
ap.add_argument("--gs", type=int, default=128)
ap.add_argument("--bits", type=int, default=4)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--default-baseline", type=int, default=1)
ap.add_argument("--layers", type=int, default=8)
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
L = _lib.LAB


def tune(v, rb=0, s=0, xb=0):
    t = _lib.GptqTuning()
    t.path, t.reserved[L.GEMM_VARIANT], t.reserved[0], t.reserved[1], t.reserved[2] = 3, v, rb, s, xb
    return t


warm = [make_layer(4096, 4096, dev, dtype=dt, seed=99)]
xw = (torch.rand(4096, 4096, device=dev) - 0.5).to(dt)
for _ in range(3):
    run(warm, xw, None, reps=200)
del warm, xw

geoms = [tuple(map(int, g.split("x"))) for g in a.geoms.split(",")]
for shp in a.shapes.split(","):
    K, N = map(int, shp.split("x"))
    for act in map(int, a.act.split(",")):
        ls = [make_layer(K, N, dev, bits=a.bits, gs=a.gs, dtype=dt, seed=i, act_order=bool(act)) for i in range(a.layers)]
        for M in map(int, a.ms.split(",")):
            x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
            best = {}
            toff = None if a.default_baseline else tune(L.VARIANT_ROWS_OFF)      # None: the planner's own choice (mid / stream64 / ... included) while rows_pays() is off
            kname = _lib.describe_plan(ls[0]._layer, M, toff).get("kernel")
            for _ in range(a.rounds):
                best["without"] = min(best.get("without", 1e9), run(ls, x, toff, reps=5))
                for rb, s in geoms:
                    if (rb == 2 and M <= 16) or (rb == 4 and M <= 48):
                        continue
                    for xb in map(int, a.xb.split(",")):
                        if not rb and xb != 2:
                            continue
                        t = tune(L.VARIANT_ROWS_ON, rb, s, 1 if xb == 1 else 0)
                        d = _lib.describe_plan(ls[0]._layer, M, t)
                        if d.get("kernel") != "rows":
                            continue
                        key = (f"{rb}x{s}" if rb else f"auto:{d['mt']}x{int(d['tiles'].split('x')[1])}sg") + ("" if xb == 2 else "/1")
                        best[key] = min(best.get(key, 1e9), run(ls, x, t, reps=5))
            w = best.pop("without")
            kb = min(best, key=best.get) if best else None
            print(f"int{a.bits} g{a.gs} {K}x{N} M={M:4d} {a.dtype} act={act} | without [{kname:9s}] {w * 1e6:7.2f} us | " +
                  " ".join(f"{k} {v * 1e6:6.2f}" for k, v in best.items()) + (f" | best {kb} {best[kb] * 1e6:6.2f} us {w / best[kb]:5.2f}x" if kb else ""), flush=True)
            del x
        del ls
