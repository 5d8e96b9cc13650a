#!/usr/bin/env python3
"""Decode (M = 1) default plans, fp16 against bf16: single Llama-7B layers and the q|k|v / gate|up launches of gptq_forward_multi, rotating HBM-cold
weights inside a hipGraph.  Usage: python tools/bf16_vs_f16.py [--m 1]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.stream_sweep import timed
from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import forward_multi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    res = {}
    for dt, name in ((torch.float16, "f16"), (torch.bfloat16, "bf16")):
        for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
            nl = max(4, min(48, (400 << 20) // (K * N // 2)))
            ls = [make_layer(K, N, dev, dtype=dt, seed=i) for i in range(nl)]
            x = (torch.rand(a.m, K, device=dev) - 0.5).to(dt)
            s, _ = timed(lambda: [q(x) for q in ls])
            res[(f"{K}x{N}", name)] = (s / nl * 1e6, _lib.describe_plan(ls[0]._layer, a.m).get("kernel"))
            del ls
            torch.cuda.empty_cache()
        for gname, K, Ns in (("q|k|v", 4096, (4096, 4096, 4096)), ("gate|up", 4096, (11008, 11008))):
            ng = max(3, (400 << 20) // (K * sum(Ns) // 2))
            groups = [[make_layer(K, n, dev, dtype=dt, seed=100 * gi + i) for i, n in enumerate(Ns)] for gi in range(ng)]
            x = (torch.rand(a.m, K, device=dev) - 0.5).to(dt)
            s, _ = timed(lambda: [forward_multi(g, x) for g in groups])
            res[(gname, name)] = (s / ng * 1e6, "multi")
            del groups
            torch.cuda.empty_cache()
    for key in ("4096x4096", "4096x11008", "11008x4096", "q|k|v", "gate|up"):
        f, b = res[(key, "f16")], res[(key, "bf16")]
        print(f"M={a.m} {key:12s} fp16 {f[0]:7.2f} us [{f[1]}]   bf16 {b[0]:7.2f} us [{b[1]}]   bf16 / fp16 = {b[0] / f[0]:.3f}")


if __name__ == "__main__":
    main()
