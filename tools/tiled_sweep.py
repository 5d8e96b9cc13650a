#!/usr/bin/env python3
"""Round 4: decode from the strip-major side copy (tuning.path = 8: gemv_tiled_kernel) against the checkpoint-layout kernels (default plan of a
layer built WITHOUT the side copy), rotating HBM-cold weights inside a hipGraph; single layers and the multi-layer launches of gptq_forward_multi.
Usage: python tools/tiled_sweep.py [--m 1] [--dtype f16] [--quick]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer, algorithmic_bytes
from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import QuantLinear, forward_multi


def timed(fn, reps=8):
    with torch.no_grad():
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        keep = fn()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / reps)
    return best, keep


def tune(waves, u, dma=False, ks=0):
    t = _lib.GptqTuning()
    t.path = 8
    t.waves = waves
    t.ksplit = ks
    t.reserved[0] = u
    return t


def plain_twin(q):
    """The same layer without the strip-major side copy (what rounds 1-3 ran)."""
    p = QuantLinear(q.bits, q.group_size, q.infeatures, q.outfeatures, False, weight_dtype=q.scales.dtype)
    p.qweight, p.qzeros, p.scales, p.g_idx = q.qweight, q.qzeros, q.scales, q.g_idx
    p = p.to(q.qweight.device)
    p.post_init(tiled=False)
    return p


CONFIGS = [(w, u, False) for (w, u) in ((16, 1), (16, 2), (16, 4), (8, 2), (8, 4), (8, 8), (4, 2), (4, 4), (4, 8), (2, 4), (2, 8))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1)
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--shapes", default="4096x4096,11008x4096,4096x11008")
    ap.add_argument("--no-multi", action="store_true")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--act", action="store_true", help="act-order (desc_act=True) layers")
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--gs", type=int, default=128)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    M = args.m
    cfgs = CONFIGS if not args.quick else [(16, 2, False), (8, 4, False), (4, 4, False)]
    bits, gs = args.bits, args.gs
    if bits != 4 or args.act:
        cfgs = [c for c in cfgs if c[1] in (2, 4)]
    for K, N in [tuple(map(int, sh.split('x'))) for sh in args.shapes.split(',')]:
        per = K * N * bits // 8
        nl = max(4, min(64, (400 << 20) // per))
        layers = [make_layer(K, N, dev, dtype=dt, seed=i, bits=bits, gs=gs, act_order=args.act) for i in range(nl)]
        twins = [plain_twin(q) for q in layers]
        x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
        ab = algorithmic_bytes(K, N, M, bits=bits, gs=gs)
        res = []
        base, ref = timed(lambda: [q(x) for q in twins])
        res.append((base / nl, "checkpoint layout (round-3 default plan)"))
        s, out = timed(lambda: [q(x) for q in layers])
        res.append((s / nl, "strip-major, default plan"))
        for waves, u, dma in cfgs:
            t = tune(waves, u, dma)
            try:
                s, out = timed(lambda: [q(x, tuning=t) for q in layers])
            except Exception as e:
                print("fail", waves, u, dma, str(e)[:100]); continue
            ok = all(torch.allclose(a.float(), b.float(), rtol=2e-2, atol=2e-2 * float(b.float().abs().max())) for a, b in zip(out[:2], ref[:2]))
            res.append((s / nl, f"strip-major waves={waves} u={u} {'' if ok else '  MISMATCH'}"))
        res.sort()
        print(f"== {K}x{N} M={M} {args.dtype}: {nl} layers, {ab} B/launch")
        for s, name in res[:16]:
            print(f"   {s*1e6:7.2f} us {ab/s/1e9:7.0f} GB/s  {name}")
        print(f"   (checkpoint layout: {base/nl*1e6:.2f} us)", flush=True)
        del layers, twins
        torch.cuda.empty_cache()
    for name, K, Ns in () if args.no_multi else (("qkv", 4096, (4096, 4096, 4096)), ("gate_up", 4096, (11008, 11008))):
        ng = max(3, (400 << 20) // (K * sum(Ns) * bits // 8))
        groups = [[make_layer(K, n, dev, dtype=dt, seed=100 * gi + i, bits=bits, gs=gs, act_order=args.act, order_seed=(7000 + gi) if args.act else None) for i, n in enumerate(Ns)] for gi in range(ng)]
        tgroups = [[plain_twin(q) for q in grp] for grp in groups]
        x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
        ab = sum(algorithmic_bytes(K, n, M, bits=bits, gs=gs) for n in Ns)
        res = []
        base, ref = timed(lambda: [forward_multi(grp, x) for grp in tgroups])
        res.append((base / ng, "checkpoint layout forward_multi (round-3 default plan)"))
        s, out = timed(lambda: [forward_multi(grp, x) for grp in groups])
        res.append((s / ng, "strip-major forward_multi, default plan"))
        for waves, u, dma in cfgs:
            t = tune(waves, u, dma)
            try:
                s, out = timed(lambda: [forward_multi(grp, x, t) for grp in groups])
            except Exception as e:
                print("fail", waves, u, dma, str(e)[:100]); continue
            ok = all(torch.allclose(a.float(), b.float(), rtol=2e-2, atol=2e-2 * float(b.float().abs().max())) for a, b in zip(out[0], ref[0]))
            res.append((s / ng, f"strip-major forward_multi waves={waves} u={u} {'' if ok else '  MISMATCH'}"))
        res.sort()
        print(f"== {name} K={K} N={Ns} M={M}: {ng} groups, {ab} B per group")
        for s, nm in res[:16]:
            print(f"   {s*1e6:7.2f} us {ab/s/1e9:7.0f} GB/s  {nm}")
        print(f"   (checkpoint layout: {base/ng*1e6:.2f} us)", flush=True)
        del groups, tgroups
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
