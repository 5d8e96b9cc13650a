#!/usr/bin/env python3
"""Context for the decode kernels' fixed cost: per-node time of a hipGraph of plain torch elementwise kernels, rotating HBM-cold
buffers, same settle protocol as bench.py.  traffic = bytes read + bytes written per node; a node of a few bytes is the launch boundary alone."""
import time
import torch

dev = torch.device("cuda:0")


def per_node(nbytes_traffic, nodes=64):
    n = max(4, nbytes_traffic // 2 // 4)                       # int32 elements in = out
    src = [torch.ones(n, dtype=torch.int32, device=dev) for _ in range(nodes)]
    dst = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(nodes)]
    for s, d in zip(src, dst):
        torch.bitwise_and(s, 1, out=d)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for s, d in zip(src, dst):
            torch.bitwise_and(s, 1, out=d)
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        g.replay()
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / (4 * nodes))
    return best


for mb in (0, 8.73, 23.46, 46.91, 93.8):
    b = int(mb * 1e6)
    t = per_node(b)
    print(f"traffic {mb:6.2f} MB per node: {t * 1e6:6.2f} us per node" + (f"  {b / t / 1e12:5.2f} TB/s" if b else "  (launch boundary)"), flush=True)
