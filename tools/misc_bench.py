#!/usr/bin/env python3
"""The secondary numbers of README.md re-measured with the current library: default plans only, rotating layers in a hipGraph.
Usage: python tools/misc_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer, algorithmic_bytes
from tools.gemv_sweep import run
from tools.stream_sweep import timed
from autogptq_amd.qlinear_mi355x import forward_multi

dev = torch.device("cuda:0")
SHAPES = ((4096, 4096), (4096, 11008), (11008, 4096))


def layers_for(K, N, **kw):
    nl = max(4, min(48, (400 << 20) // (K * N // 2)))
    return [make_layer(K, N, dev, seed=i, **kw) for i in range(nl)]


def line(tag, vals):
    print(f"{tag:46s} " + " / ".join(f"{v * 1e6:6.2f}" for v in vals) + " us", flush=True)


for M in (2, 4, 8):
    vals = []
    for K, N in SHAPES:
        ls = layers_for(K, N)
        vals.append(run(ls, (torch.rand(M, K, device=dev) - 0.5).half(), None))
        del ls
    line(f"fp16 plain M={M} (4096^2, 4096x11008, 11008x4096)", vals)
for tag, kw, dt in (("act-order fp16 M=1", dict(act_order=True), torch.float16), ("bf16 plain M=1", dict(dtype=torch.bfloat16), torch.bfloat16),
                    ("bf16 act-order M=1", dict(act_order=True, dtype=torch.bfloat16), torch.bfloat16)):
    vals = []
    for K, N in SHAPES:
        ls = layers_for(K, N, **kw)
        vals.append(run(ls, (torch.rand(1, K, device=dev) - 0.5).to(dt), None))
        del ls
    line(tag, vals)
for dt, name in ((torch.float16, "fp16"), (torch.bfloat16, "bf16")):
    for M in (1, 2, 4, 8, 16, 32, 64):
        vals = []
        for K, Ns in ((4096, (4096,) * 3), (4096, (11008,) * 2)):
            ng = max(3, (400 << 20) // (K * sum(Ns) // 2))
            groups = [[make_layer(K, n, dev, dtype=dt, seed=100 * gi + i) for i, n in enumerate(Ns)] for gi in range(ng)]
            x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
            s, _ = timed(lambda: [forward_multi(g, x) for g in groups])
            vals.append(s / ng)
            del groups
        line(f"forward_multi {name} M={M} (q|k|v, gate|up)", vals)
vals = []
for K, N in ((8192, 1024), (8192, 3584), (28672, 1024), (3584, 8192)):
    ls = layers_for(K, N)
    vals.append(run(ls, (torch.rand(1, K, device=dev) - 0.5).half(), None))
    del ls
line("70B TP=8 shards M=1 (8192x1024, 8192x3584, 28672x1024, 3584x8192)", vals)
for M in (1, 4):
    vals = []
    for K, N in ((5120, 5120), (5120, 13824), (13824, 5120), (8192, 8192), (8192, 28672), (28672, 8192)):
        ls = layers_for(K, N)
        vals.append(run(ls, (torch.rand(M, K, device=dev) - 0.5).half(), None))
        del ls
    line(f"13B / 70B layers M={M} (5120^2, 5120x13824, 13824x5120, 8192^2, 8192x28672, 28672x8192)", vals)
for M in (9, 16, 32, 64):
    vals = []
    for K, N in SHAPES:
        ls = layers_for(K, N)
        vals.append(run(ls, (torch.rand(M, K, device=dev) - 0.5).half(), None))
        del ls
    line(f"batched decode M={M}", vals)
