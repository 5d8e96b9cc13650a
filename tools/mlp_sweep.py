#!/usr/bin/env python3
"""gptq_mlp_forward (gate, up, down in ONE C-ABI call) against the same block composed from its parts -- forward_multi([gate, up]) + torch SiLU * mul + down -- across model
families and row counts, plain and act-order; HBM-cold rotating blocks in a hipGraph.  The call losing to its own parts = a rule of the MLP path to fix.
usage: python tools/mlp_sweep.py [--act] [--ms 1,4,16,64,256,2048]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from autogptq_amd.qlinear_mi355x import forward_multi, mlp_forward  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ms", default="1,2,4,8,16,32,64,128,256,512,2048")
ap.add_argument("--act", action="store_true")
args = ap.parse_args()
dev = torch.device("cuda:0")
FAMILIES = [("7B", 4096, 11008), ("13B", 5120, 13824), ("30B", 6656, 17920), ("8B", 4096, 14336), ("70B TP8", 8192, 3584)]


def timed(fn, nblk, reps=5):
    with torch.no_grad():
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        outs = fn()
    bench.settle(g, dev)
    _, evt = bench.time_graph(g, reps, dev)
    del g, outs
    return evt / (reps * nblk) * 1e6


for name, H, I in FAMILIES:
    nb = max(3, min(12, -(-(320 << 20) // (3 * H * I // 2))))
    blocks = [(bench.make_layer(H, I, dev, act_order=args.act, seed=9800 + 4 * i, order_seed=70_000 + i), bench.make_layer(H, I, dev, act_order=args.act, seed=9801 + 4 * i, order_seed=70_000 + i),
               bench.make_layer(I, H, dev, act_order=args.act, seed=9802 + 4 * i)) for i in range(nb)]
    for M in (int(m) for m in args.ms.split(",")):
        x = (torch.rand(M, H, device=dev) - 0.5).half()

        def composed():
            outs = []
            for g_, u_, d_ in blocks:
                a, b = forward_multi([g_, u_], x)
                outs.append(d_(torch.nn.functional.silu(a) * b))
            return outs
        t_call = timed(lambda: [mlp_forward(g_, u_, d_, x) for g_, u_, d_ in blocks], nb)
        t_parts = timed(composed, nb)
        flag = "   <-- the MLP call LOSES to its parts" if t_call > 1.03 * t_parts else ""
        print(f"{name:8s} {H}->{I}->{H} act={int(args.act)} M={M:5d}: mlp_forward {t_call:8.2f} us | composed {t_parts:8.2f} us{flag}", flush=True)
    del blocks
    torch.cuda.empty_cache()
