#!/usr/bin/env python3
"""Tiled MFMA GEMM variants (tuning.reserved[3]) on the prefill shapes: rotating layers inside a hipGraph, HIP events.
Usage: python tools/gemm_sweep.py [--variants 0,8,9] [--dtype f16]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from autogptq_amd import _lib
from tools.gemv_sweep import run


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,8,9")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--cases", default="4096x4096x2048a,4096x11008x2048a,11008x4096x2048a,4096x4096x4096,4096x4096x2048,4096x11008x512a,4096x4096x1024a")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[args.dtype]
    for case in args.cases.split(","):
        act = case.endswith("a")
        K, N, M = map(int, case.rstrip("a").split("x"))
        layers = [make_layer(K, N, dev, act_order=act, dtype=dt, seed=i) for i in range(6)]
        x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
        row, ref = [], None
        for v in map(int, args.variants.split(",")):
            t = _lib.GptqTuning()
            t.path = 3
            t.reserved[3] = v
            try:
                s = run(layers, x, t, reps=4)
                with torch.no_grad():
                    y = layers[0](x, tuning=t)
                if ref is None:
                    ref = y
                ok = bool(torch.allclose(y.float(), ref.float(), rtol=1e-2, atol=1e-2 * float(ref.float().abs().max())))
                plan = _lib.describe_plan(layers[0]._layer, M, t)
                row.append(f"v{v} {s * 1e6:7.1f} us {2 * M * K * N / s / 1e12:6.0f} TF [{plan['kernel']} kg={plan.get('kg')} {plan.get('tiles')}]{'' if ok else ' MISMATCH'}")
            except Exception as e:
                row.append(f"v{v} fail {str(e)[:60]}")
        print(f"{K}x{N} M={M} act={int(act)}: " + " | ".join(row), flush=True)
        del layers
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
