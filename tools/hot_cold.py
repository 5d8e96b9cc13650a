#!/usr/bin/env python3
"""Decode launches on weights that are HBM-cold (rotating > 256 MiB of layers) against weights resident in the 256 MiB Infinity Cache (the same few
layers over and over): what would a prefetch of the next layer's weights into the Infinity Cache buy?
Usage: python tools/hot_cold.py [--m 1]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.stream_sweep import timed
from autogptq_amd.qlinear_mi355x import forward_multi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, K, Ns in (("4096x4096", 4096, (4096,)), ("4096x11008", 4096, (11008,)), ("11008x4096", 11008, (4096,)), ("q|k|v", 4096, (4096, 4096, 4096)),
                        ("gate|up", 4096, (11008, 11008))):
        per = K * sum(Ns) // 2
        ncold = max(6, (640 << 20) // per)
        groups = [[make_layer(K, n, dev, seed=100 * gi + i) for i, n in enumerate(Ns)] for gi in range(ncold)]
        x = (torch.rand(a.m, K, device=dev) - 0.5).half()
        call = (lambda g: g[0](x)) if len(Ns) == 1 else (lambda g: forward_multi(g, x))
        cold, _ = timed(lambda: [call(g) for g in groups])
        nhot = max(2, min(ncold, (96 << 20) // per))            # <= 96 MiB: stays in the Infinity Cache, larger than one XCD's L2
        reps = -(-ncold // nhot)
        hot, _ = timed(lambda: [call(g) for _ in range(reps) for g in groups[:nhot]])
        one, _ = timed(lambda: [call(groups[0]) for _ in range(ncold)])
        print(f"M={a.m} {name:12s} cold ({ncold} layers, {ncold * per >> 20} MiB) {cold / ncold * 1e6:6.2f} us | Infinity-Cache hot ({nhot} layers, {nhot * per >> 20} MiB) "
              f"{hot / (reps * nhot) * 1e6:6.2f} us | one layer repeated {one / ncold * 1e6:6.2f} us")
        del groups
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
