#!/usr/bin/env python3
"""act-order decode (M = 1, fp16/bf16): default plan against forced rows-per-lane / waves on the three Llama-7B shapes.
Usage: python tools/act_ab.py  (GPTQ_MI355X_LIB selects a variant library)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib

dev = torch.device("cuda:0")
for dt in (torch.float16, torch.bfloat16):
    for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
        nl = max(4, min(48, (400 << 20) // (K * N // 2)))
        ls = [make_layer(K, N, dev, act_order=True, dtype=dt, seed=i) for i in range(nl)]
        x = (torch.rand(1, K, device=dev) - 0.5).to(dt)
        res = []
        for waves in (0, 16, 8):
            for u in (0, 1, 2, 4, 8):
                if (waves == 0) != (u == 0):
                    continue
                t = _lib.GptqTuning()
                t.waves = waves
                t.reserved[0] = u
                try:
                    res.append(f"w{waves}u{u}={run(ls, x, t) * 1e6:.2f}")
                except Exception as e:
                    res.append(f"w{waves}u{u}=fail")
        print(str(dt)[6:], f"{K}x{N}", _lib.describe_plan(ls[0]._layer, 1).get("u"), " ".join(res), flush=True)
        del ls
