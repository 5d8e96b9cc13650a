#!/usr/bin/env python3
"""q|k|v and gate|up at 5 .. 128 rows through gptq_forward_multi, rotating HBM-cold weights in a hipGraph.  Run twice: with GPTQ_LAB_NO_ROWS=1 (the planner
before csrc/gemm_rows.hip) and without.  Usage: python tools/rows_multi_ab.py [--ms 8,16,32,64,128]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.stream_sweep import timed
from autogptq_amd.qlinear_mi355x import forward_multi

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="8,16,32,64,128")
a = ap.parse_args()
dev = torch.device("cuda:0")
tag = "old default" if os.environ.get("GPTQ_LAB_NO_ROWS") else "new default"
for gname, K, Ns in (("q|k|v", 4096, (4096, 4096, 4096)), ("gate|up", 4096, (11008, 11008)), ("q|k|v 13B", 5120, (5120, 5120, 5120)), ("gate|up 13B", 5120, (13824, 13824))):
    ng = max(3, (400 << 20) // (K * sum(Ns) // 2))
    groups = [[make_layer(K, n, dev, seed=100 * gi + i) for i, n in enumerate(Ns)] for gi in range(ng)]
    for M in map(int, a.ms.split(",")):
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        one, _ = timed(lambda: [forward_multi(g, x) for g in groups])
        sep, _ = timed(lambda: [[q(x) for q in g] for g in groups])
        print(f"[{tag}] {gname:12s} M={M:3d}: forward_multi {one / ng * 1e6:7.2f} us | layer by layer {sep / ng * 1e6:7.2f} us", flush=True)
    del groups
    torch.cuda.empty_cache()
