#!/usr/bin/env python3
"""Times the batched-decode kernel (default plan) at M = 32 / 64 on the three Llama-7B shapes -- run once per ablation build
(tools/ab_build.sh with -DGPTQ_S64_ABL=n copies of gemm.hip: 1 no x loads, 2 no bpermute / perm of the x fragments, 4 no dequant math, 8 no MFMAs;
GPTQ_MI355X_LIB selects the library).  Ablation builds give wrong results by design."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib

dev = torch.device("cuda:0")
out = []
for K, N in ((4096, 11008), (11008, 4096)):
    nl = max(4, min(32, (400 << 20) // (K * N // 2)))
    ls = [make_layer(K, N, dev, seed=i) for i in range(nl)]
    for M in (32, 64):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        out.append(f"{K}x{N} M={M} [{_lib.describe_plan(ls[0]._layer, M).get('kernel')}]: {run(ls, x, None) * 1e6:6.2f}")
    del ls
    torch.cuda.empty_cache()
print(os.environ.get("GPTQ_MI355X_LIB", "in-tree").split("/")[-1], " | ".join(out), flush=True)
