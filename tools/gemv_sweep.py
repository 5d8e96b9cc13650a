#!/usr/bin/env python3
"""Sweep GEMV launch shapes (strip width, waves, K split, path) on rotating, HBM-cold weights.
Usage: python tools/gemv_sweep.py [--m 1] [--shapes 4096x4096,4096x11008,11008x4096]"""
import argparse, itertools, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer, algorithmic_bytes
from autogptq_amd import _lib


def run(layers, x, tune, reps=5):
    dev = x.device
    with torch.no_grad():
        for q in layers[:2]:
            q(x, tuning=tune)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        outs = [q(x, tuning=tune) for q in layers]
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (reps * len(layers))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--gs", type=int, default=128)
    ap.add_argument("--full", action="store_true")
    ap.add_argument("--heuristic-only", action="store_true")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--act", action="store_true", help="act-order layers (desc_act=True)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for shp in args.shapes.split(","):
        K, N = map(int, shp.split("x"))
        per = K * N * args.bits // 8
        nl = max(4, min(64, (640 << 20) // per))           # > 256 MiB of distinct weights
        dt = torch.float16 if args.dtype == 'f16' else torch.bfloat16
        layers = [make_layer(K, N, dev, bits=args.bits, gs=args.gs, act_order=args.act, dtype=dt, seed=i) for i in range(nl)]
        x = (torch.rand(args.m, K, device=dev) - 0.5).to(dt)
        ab = algorithmic_bytes(K, N, args.m, bits=args.bits, gs=args.gs, act_order=args.act)
        cfgs = [dict()]  # heuristic
        lns = (4, 8, 16, 64)
        for ln in lns:
            for waves in ((4, 8, 16) if not args.full else (2, 4, 8, 16)):
                for ks in ((1, 2, 4, 8) if ln >= 16 else (1, 2)):
                    cfgs.append(dict(lanes_n=ln, waves=waves, ksplit=ks, path=2 if args.bits == 4 else 1))
                    if args.bits == 4:
                        cfgs.append(dict(lanes_n=ln, waves=waves, ksplit=ks, path=4))
                        cfgs.append(dict(lanes_n=ln, waves=waves, ksplit=ks, path=5))
                    elif ln == 4 and not args.act:
                        cfgs.append(dict(lanes_n=ln, waves=waves, ksplit=ks, path=5))      # matrix-core kernel with field extraction
        cfgs.append(dict(path=1))
        if args.heuristic_only:
            cfgs = [dict(), dict(path=1)] + ([dict(path=2)] if args.bits == 4 and args.dtype == 'f16' else [])
        res = []
        for c in cfgs:
            t = _lib.GptqTuning()
            for k, v in c.items():
                setattr(t, k, v)
            try:
                s = run(layers, x, t)
            except Exception as e:
                print("fail", c, e); continue
            res.append((s, c))
        res.sort(key=lambda r: r[0])
        print(f"== {K}x{N} M={args.m} bits={args.bits}: {nl} layers, {ab} B/launch")
        for s, c in res[:24]:
            print(f"   {s*1e6:8.2f} us  {ab/s/1e9:8.1f} GB/s  {c}")
        hs = [r for r in res if r[1] == {}]
        print(f"   heuristic: {hs[0][0]*1e6:.2f} us   generic: {[r for r in res if r[1]==dict(path=1)][0][0]*1e6:.2f} us")
        del layers
        torch.cuda.empty_cache()

if __name__ == "__main__":
    main()
