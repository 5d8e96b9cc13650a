#!/usr/bin/env python3
"""2-bit prefill (M = 2048) on the tiled kernel.  Usage: python tools/int2_prefill.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib
dev = torch.device("cuda:0")
for bits, gs in ((2, 64), (2, 128), (4, 128)):
    for K, N in ((4096, 4096), (4096, 11008)):
        ls = [make_layer(K, N, dev, bits=bits, gs=gs, seed=i) for i in range(6)]
        x = (torch.rand(2048, K, device=dev) - 0.5).half()
        t = run(ls, x, None)
        print(f"int{bits} g{gs} {K}x{N} M=2048: {t * 1e6:7.1f} us = {2 * 2048 * K * N / t / 1e12:6.1f} TFLOP/s  [{_lib.describe_plan(ls[0]._layer, 2048).get('kernel')} bk={_lib.describe_plan(ls[0]._layer, 2048).get('bk')}]", flush=True)
        del ls
        torch.cuda.empty_cache()
