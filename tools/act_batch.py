import os, sys
sys.path.insert(0, os.getcwd())
import torch
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib
dev = torch.device("cuda:0")
for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
    nl = max(4, min(32, (400 << 20) // (K * N // 2)))
    la = [make_layer(K, N, dev, act_order=True, seed=i) for i in range(nl)]
    lp = [make_layer(K, N, dev, seed=i) for i in range(nl)]
    out = []
    for M in (1, 2, 3, 4, 5, 8, 16, 32):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        out.append(f"M={M}: act {run(la, x, None) * 1e6:.2f} [{_lib.describe_plan(la[0]._layer, M).get('kernel')}] plain {run(lp, x, None) * 1e6:.2f}")
    print(f"{K}x{N}  " + " | ".join(out), flush=True)
    del la, lp
