#!/usr/bin/env python3
"""3- / 8-bit layers at 5..128 rows (default plans) next to the 4-bit layer of the same shape: what a gemm_mid_kernel for the other packings would be worth.
Usage: python tools/nonq4_batched.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib
dev = torch.device("cuda:0")
for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
    for bits, gs in ((4, 128), (4, 32), (8, 32), (3, 32), (2, 64)):
        n = max(4, min(24, (320 << 20) // (K * N * bits // 8)))
        ls = [make_layer(K, N, dev, bits=bits, gs=gs, seed=i) for i in range(n)]
        out = []
        for M in (5, 8, 16, 32, 64, 128):
            x = (torch.rand(M, K, device=dev) - 0.5).half()
            t = run(ls, x, None)
            extra = ""
            if bits in (8, 3, 2):
                tn = _lib.GptqTuning(); tn.path = 3; tn.reserved[2] = 5
                try:
                    extra = f" mid {run(ls, x, tn) * 1e6:6.2f}"
                except Exception as e:
                    extra = " mid n/a"
            out.append(f"M={M}: {t * 1e6:6.2f} us [{_lib.describe_plan(ls[0]._layer, M).get('kernel')}]{extra}")
        print(f"{K}x{N} int{bits} g{gs}: " + " | ".join(out), flush=True)
        del ls
        torch.cuda.empty_cache()
