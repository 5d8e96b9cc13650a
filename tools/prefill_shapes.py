#!/usr/bin/env python3
"""Prefill (default plans) on layer shapes beyond Llama-7B: us per call and TFLOP/s at M = 512 / 2048, plain and act-order.
Usage: python tools/prefill_shapes.py [--ms 512,2048]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="512,2048")
ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096,5120x5120,5120x13824,13824x5120,8192x8192,8192x28672,28672x8192,8192x1024,8192x3584,28672x1024")
a = ap.parse_args()
dev = torch.device("cuda:0")
for shp in a.shapes.split(","):
    K, N = map(int, shp.split("x"))
    for act in (False, True):
        nl = max(2, min(8, (300 << 20) // (K * N // 2)))
        ls = [make_layer(K, N, dev, act_order=act, seed=i) for i in range(nl)]
        out = []
        for M in map(int, a.ms.split(",")):
            x = (torch.rand(M, K, device=dev) - 0.5).half()
            s = run(ls, x, None, reps=3)
            d = _lib.describe_plan(ls[0]._layer, M)
            out.append(f"M={M}: {s * 1e6:8.1f} us {2 * M * K * N / s / 1e12:6.0f} TF [{d.get('kernel')} ks={d.get('ksplit')} kg={d.get('kg')} tiles={d.get('tiles')}]")
        print(f"{K}x{N} act={int(act)}  " + "   ".join(out), flush=True)
        del ls
