#!/usr/bin/env python3
"""Round 6: the whole-K panel kernel (csrc/gemm_panel.hip, tuning.reserved[3] = 52, reserved[0] = 20 + NT) against the planner's choice
without it (53), layer call (incl. the x permute of act-order layers) on rotating layers in a hipGraph, interleaved rounds, minimum per variant; the first
round of every configuration also compares every output of the two forms.
Usage: python tools/panel_ab.py [--ms 128,256,512] [--shapes 4096x4096,...] [--geoms 0,21,22,23,24] [--act 0,1] [--dtype f16]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="128,192,256,384,512,768")
ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
ap.add_argument("--geoms", default="0,21,22,23,24", help="20 + NT; 0 = the kernel's own planner")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--act", default="0")
ap.add_argument("--gs", type=int, default=128)
ap.add_argument("--bits", type=int, default=4)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--layers", type=int, default=0, help="rotating layers per shape; 0 = enough for > 512 MiB of packed weights (HBM-cold, as bench.py measures), at most 64")
ap.add_argument("--check", type=int, default=1)
ap.add_argument("--default-baseline", type=int, default=1)
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
L = _lib.LAB


def tune(v, g=0, kp=0):
    t = _lib.GptqTuning()
    t.path, t.reserved[L.GEMM_VARIANT], t.reserved[0], t.reserved[1] = 3, v, g, kp
    return t


warm = [make_layer(4096, 4096, dev, dtype=dt, seed=99)]
xw = (torch.rand(4096, 4096, device=dev) - 0.5).to(dt)
for _ in range(3):
    run(warm, xw, None, reps=100)
del warm, xw

geoms = [(int(g.split("x")[0]), 0) for g in a.geoms.split(",")]
for shp in a.shapes.split(","):
    K, N = map(int, shp.split("x"))
    for act in map(int, a.act.split(",")):
        nl = a.layers or max(4, min(64, (640 << 20) // (K * N * a.bits // 8)))
        ls = [make_layer(K, N, dev, bits=a.bits, gs=a.gs, dtype=dt, seed=i, act_order=bool(act)) for i in range(nl)]
        for M in map(int, a.ms.split(",")):
            x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
            best, bad = {}, {}
            toff = None if a.default_baseline else tune(L.VARIANT_PANEL_OFF)      # None: the planner's own choice (run with GPTQ_LAB_NO_PANEL=1 once panel_pays() is on)
            kname = _lib.describe_plan(ls[0]._layer, M, toff).get("kernel")
            with torch.no_grad():
                yref = ls[0](x, tuning=toff).float()
            tol = (1e-3 if a.dtype == "f16" else 8e-3) * 2
            for rnd in range(a.rounds):
                best["without"] = min(best.get("without", 1e9), run(ls, x, toff, reps=5))
                for g, kp in geoms:
                    t = tune(L.VARIANT_PANEL_ON, g, kp)
                    d = _lib.describe_plan(ls[0]._layer, M, t)
                    if d.get("kernel") != "panel":
                        continue
                    key = f"{g}" if g else f"auto:{d['mt']}x{d['tiles']}w{d['waves']}"
                    if rnd == 0 and a.check:
                        with torch.no_grad():
                            y = ls[0](x, tuning=t).float()
                        nbad = int(((y - yref).abs() > tol * yref.abs().max() + tol * yref.abs()).sum())
                        if nbad:
                            bad[key] = nbad
                    best[key] = min(best.get(key, 1e9), run(ls, x, t, reps=5))
            w = best.pop("without")
            kb = min(best, key=best.get) if best else None
            tf = 2.0 * M * K * N / 1e12
            print(f"int{a.bits} g{a.gs} {K}x{N} M={M:4d} {a.dtype} act={act} | without [{kname:9s}] {w * 1e6:7.2f} us | " +
                  " ".join(f"{k} {v * 1e6:6.2f}" for k, v in best.items()) +
                  (f" | best {kb} {best[kb] * 1e6:6.2f} us {tf / best[kb]:5.0f} TF {w / best[kb]:5.2f}x" if kb else "") +
                  (f" | MISMATCH {bad}" if bad else ""), flush=True)
            del x
        del ls
