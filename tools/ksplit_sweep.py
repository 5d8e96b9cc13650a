#!/usr/bin/env python3
"""Tiled MFMA GEMM at mid M: K split (tuning.ksplit with path = 3) sweep per shape.  Usage: python tools/ksplit_sweep.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from autogptq_amd import _lib
from tools.gemv_sweep import run

dev = torch.device("cuda:0")
for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
    layers = [make_layer(K, N, dev, seed=i) for i in range(8)]
    for M in (65, 96, 128, 192, 256, 384, 512):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        row = []
        for ks in (0, 1, 2, 3, 4, 6, 8):
            t = _lib.GptqTuning()
            t.path = 3
            t.ksplit = ks
            t.reserved[2] = 2          # tiled kernel
            try:
                s = run(layers, x, t, reps=4)
                p = _lib.describe_plan(layers[0]._layer, M, t)
                row.append(f"ks{ks}->{p['ksplit']}/kg{p['kg']}: {s * 1e6:6.1f}")
            except Exception as e:
                row.append(f"ks{ks}: fail")
        print(f"{K}x{N} M={M:3d}: " + " | ".join(row), flush=True)
    del layers
    torch.cuda.empty_cache()
