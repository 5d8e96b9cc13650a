#!/usr/bin/env python3
"""Round 5: batched decode on the decode copy (csrc/gemm_strips.hip, tuning.reserved[2] = 6) against the planner's choice without it (7: strip16 / stream64 /
mid / skinny on the checkpoint rows), rotating HBM-cold layers in a hipGraph.  Usage: python tools/strips_ab.py [--ms 5,8,16,...] [--shapes ...] [--act 0,1]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer, algorithmic_bytes
from tools.gemv_sweep import run
from autogptq_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="5,8,16,24,32,48,64")
ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--act", default="0")
ap.add_argument("--ks", default="0", help="K slices to try for the new kernel (0 = its planner)")
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float16 if a.dtype == "f16" else torch.bfloat16


def tune(v, ks=0):
    t = _lib.GptqTuning()
    t.path, t.reserved[2], t.ksplit = (3 if v == 6 else 0), v, ks      # 7: the planner's own choice (GEMV or GEMM) with the new kernel switched off
    return t


for shp in a.shapes.split(","):
    K, N = map(int, shp.split("x"))
    per = K * N // 2
    nl = max(4, min(48, (640 << 20) // (3 * per)))
    for act in map(int, a.act.split(",")):
        ls = [make_layer(K, N, dev, dtype=dt, seed=i, act_order=bool(act)) for i in range(nl)]
        for M in map(int, a.ms.split(",")):
            x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
            best, names = {}, {}
            for _ in range(a.rounds):
                for name, v, ks in [("without", 7, 0)] + [(f"strips ks={k}" if k != "0" else "strips", 6, int(k)) for k in a.ks.split(",")]:
                    t = tune(v, ks)
                    try:
                        d = _lib.describe_plan(ls[0]._layer, M, t)
                    except Exception:
                        continue
                    names[name] = f"{d.get('kernel')} ks={d.get('ksplit')}"
                    s = run(ls, x, t, reps=5)
                    best[name] = min(best.get(name, 1e9), s)
            w = best.pop("without")
            ab = algorithmic_bytes(K, N, M)
            print(f"{K}x{N} M={M:3d} {a.dtype} act={act} | without [{names['without']:14s}] {w * 1e6:7.2f} us | " +
                  " | ".join(f"{k} [{names[k]}] {s * 1e6:7.2f} us {ab / s / 1e12:5.2f} TB/s {w / s:5.2f}x" for k, s in best.items()), flush=True)
        del ls
