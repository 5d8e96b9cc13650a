#!/usr/bin/env python3
"""bf16 decode of 2- / 3- / 8-bit layers: packed magic-number decode + one conversion per pair (default) against the field-by-field form
(tuning.path = 5, reserved[1] = 1).  Usage: python tools/bf16_magic_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib
dev = torch.device("cuda:0")
for bits, gs in ((8, 32), (3, 32), (2, 64)):
    for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
        n = max(4, min(24, (320 << 20) // (K * N * bits // 8)))
        ls = [make_layer(K, N, dev, bits=bits, gs=gs, dtype=torch.bfloat16, seed=i) for i in range(n)]
        out = []
        for M in (1, 4):
            x = (torch.rand(M, K, device=dev) - 0.5).bfloat16()
            t = _lib.GptqTuning(); t.path = 5; t.reserved[1] = 1
            a, b = run(ls, x, None), run(ls, x, t)
            out.append(f"M={M}: {a * 1e6:6.2f} us [{_lib.describe_plan(ls[0]._layer, M).get('deq')}] | field by field {b * 1e6:6.2f}")
        print(f"bf16 int{bits} g{gs} {K}x{N}: " + "   ".join(out), flush=True)
        del ls
        torch.cuda.empty_cache()
