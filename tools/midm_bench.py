#!/usr/bin/env python3
"""Batched-decode sizes (8 < M <= 64): 16-column-strip kernel vs the 64-column skinny kernel vs the tiled kernel,
rotating > 256 MiB of weights inside one hipGraph.  Usage: python tools/midm_bench.py [--dtype bf16] [--act]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer, algorithmic_bytes
from autogptq_amd import _lib
from tools.gemv_sweep import run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
    ap.add_argument("--ms", default="9,16,32,48,64")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--act", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    for shp in args.shapes.split(","):
        K, N = map(int, shp.split("x"))
        nl = max(4, min(64, (640 << 20) // (K * N // 2)))
        layers = [make_layer(K, N, dev, act_order=args.act, dtype=dt, seed=i) for i in range(nl)]
        for M in map(int, args.ms.split(",")):
            x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
            ab = algorithmic_bytes(K, N, M, act_order=args.act)
            row = []
            for name, force in (("strip16", 3), ("skinny64", 1), ("tiled", 2), ("auto", 0)):
                t = _lib.GptqTuning()
                t.path = 3
                t.reserved[2] = force
                s = run(layers, x, t)
                row.append(f"{name} {s * 1e6:7.2f} us")
            best = min(float(r.split()[1]) for r in row)
            print(f"{K}x{N} M={M:3d}: " + " | ".join(row) + f"   ({ab / best / 1e3:.0f} GB/s best)", flush=True)
        del layers
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
