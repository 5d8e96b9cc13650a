#!/usr/bin/env python3
"""Streamed GEMV (tuning.path = 6) vs the register kernel on rotating, HBM-cold weights inside a hipGraph: single layers over
(strip width, waves, rows per lane, K split) and the multi-layer launches (q/k/v, gate/up) of gptq_forward_multi.
Usage: python tools/stream_sweep.py [--m 1] [--dtype f16]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer, algorithmic_bytes
from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import forward_multi


def timed(fn, reps=6):
    with torch.no_grad():
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        keep = fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, keep


def tune(**kw):
    t = _lib.GptqTuning()
    u = kw.pop("u", 0)
    depth = kw.pop("depth", 0)
    for k, v in kw.items():
        setattr(t, k, v)
    t.reserved[0] = u
    t.reserved[1] = depth
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
    ap.add_argument("--no-multi", action="store_true")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--gs", type=int, default=128)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    M = args.m
    for K, N in [tuple(map(int, sh.split('x'))) for sh in args.shapes.split(',')]:
        per = K * N * args.bits // 8
        nl = max(4, min(64, (512 << 20) // per))
        layers = [make_layer(K, N, dev, bits=args.bits, gs=args.gs, dtype=dt, seed=i) for i in range(nl)]
        x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
        ab = algorithmic_bytes(K, N, M, bits=args.bits, gs=args.gs)
        res = []
        base, ref = timed(lambda: [q(x) for q in layers])
        res.append((base / nl, "register kernel (default plan)"))
        rows = K // (32 // args.bits if args.bits != 3 else 32)
        for ln in (4, 8, 16):
            wr = 64 // ln
            for waves, u in ((2, 8), (4, 2), (4, 4), (4, 8), (8, 2), (8, 4), (8, 8), (16, 2), (16, 4), (16, 8), (4, 1), (8, 1), (16, 1)):
                for ks in ((1, 2, 4, 8) if N < 4096 else ((1,) if ln == 4 else (1, 2, 4))):
                    if N // (4 * ln) * ks < 96 or N // (4 * ln) * ks > 1024:
                        continue
                    passes = -(-(rows // ks) // (waves * wr * u))
                    t = tune(path=6, lanes_n=ln, waves=waves, ksplit=ks, u=u)
                    try:
                        s, out = timed(lambda: [q(x, tuning=t) for q in layers])
                    except Exception as e:
                        print("fail", ln, waves, u, ks, str(e)[:80]); continue
                    ok = all(torch.allclose(a.float(), b.float(), rtol=2e-2, atol=2e-2 * float(b.float().abs().max())) for a, b in zip(out[:2], ref[:2]))
                    res.append((s / nl, f"stream ln={ln} waves={waves} u={u} ksplit={ks} passes={passes}{'' if ok else '  MISMATCH'}"))
        res.sort()
        print(f"== {K}x{N} M={M} {args.dtype}: {nl} layers, {ab} B/launch")
        for s, name in res[:18]:
            print(f"   {s*1e6:7.2f} us {ab/s/1e9:7.0f} GB/s  {name}")
        print(f"   (register kernel: {base/nl*1e6:.2f} us)")
        del layers
        torch.cuda.empty_cache()
    # multi-layer launches: groups of layers sharing x
    for name, K, Ns in () if args.no_multi else (("qkv", 4096, (4096, 4096, 4096)), ("gate_up", 4096, (11008, 11008))):
        ng = max(3, (512 << 20) // (K * sum(Ns) * args.bits // 8))
        groups = [[make_layer(K, n, dev, bits=args.bits, gs=args.gs, dtype=dt, seed=100 * gi + i) for i, n in enumerate(Ns)] for gi in range(ng)]
        x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
        ab = sum(algorithmic_bytes(K, n, M, bits=args.bits, gs=args.gs) for n in Ns)
        res = []
        base, ref = timed(lambda: [[q(x) for q in grp] for grp in groups])
        res.append((base / ng, "separate launches (register kernel)"))
        for ln in (4, 8, 16):
            for waves, u in ((2, 8), (4, 2), (4, 4), (4, 8), (8, 2), (8, 4), (16, 2), (16, 4), (8, 8), (16, 8), (4, 1), (8, 1), (16, 1)):
              for ks in ((1,) if ln == 4 else (1, 2, 4)):
                t = tune(path=6, lanes_n=ln, waves=waves, ksplit=ks, u=u)
                try:
                    s, out = timed(lambda: [forward_multi(grp, x, t) for grp in groups])
                except Exception as e:
                    continue
                ok = all(torch.allclose(a.float(), b.float(), rtol=2e-2, atol=2e-2 * float(b.float().abs().max())) for a, b in zip(out[0], ref[0]))
                res.append((s / ng, f"forward_multi ln={ln} waves={waves} u={u} ksplit={ks}{'' if ok else '  MISMATCH'}"))
        s, out = timed(lambda: [forward_multi(grp, x) for grp in groups])
        res.append((s / ng, "forward_multi (default plan)"))
        res.sort()
        print(f"== {name} K={K} N={Ns} M={M}: {ng} groups, {ab} B per group")
        for s, nm in res[:14]:
            print(f"   {s*1e6:7.2f} us {ab/s/1e9:7.0f} GB/s  {nm}")
        del groups
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
