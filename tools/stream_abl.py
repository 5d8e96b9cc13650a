#!/usr/bin/env python3
"""Times a few fixed streamed-GEMV launches (rotating layers, hipGraph) -- run once per ablation build (tools/ab_build.sh with
-DGPTQ_STREAM_ABL=n copies of gemv.hip; GPTQ_MI355X_LIB selects the library).  Ablation builds give wrong results by design."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer, algorithmic_bytes
from tools.stream_sweep import timed, tune
from autogptq_amd.qlinear_mi355x import forward_multi

dev = torch.device("cuda:0")
out = []
for name, K, N, cfg in (("down ", 11008, 4096, dict(lanes_n=4, waves=8, u=2, ksplit=1)), ("down ", 11008, 4096, dict(lanes_n=16, waves=16, u=2, ksplit=4)),
                        ("down ", 11008, 4096, dict(lanes_n=4, waves=16, u=8, ksplit=1)), ("o    ", 4096, 4096, dict(lanes_n=4, waves=8, u=2, ksplit=1)),
                        ("gate ", 4096, 11008, dict(lanes_n=16, waves=16, u=8, ksplit=1))):
    nl = max(4, min(64, (512 << 20) // (K * N // 2)))
    layers = [make_layer(K, N, dev, seed=i) for i in range(nl)]
    x = (torch.rand(1, K, device=dev) - 0.5).half()
    t = tune(path=6, **cfg)
    s, _ = timed(lambda: [q(x, tuning=t) for q in layers])
    out.append(f"{name}{K}x{N} {cfg['lanes_n']:2d}/{cfg['waves']:2d}/{cfg['u']}/ks{cfg['ksplit']}: {s / nl * 1e6:6.2f}")
    del layers
    torch.cuda.empty_cache()
for name, K, Ns, cfg in (("qkv  ", 4096, (4096,) * 3, dict(lanes_n=16, waves=16, u=8, ksplit=1)), ("gateup", 4096, (11008,) * 2, dict(lanes_n=8, waves=8, u=2, ksplit=1))):
    ng = max(3, (512 << 20) // (K * sum(Ns) // 2))
    groups = [[make_layer(K, n, dev, seed=100 * gi + i) for i, n in enumerate(Ns)] for gi in range(ng)]
    x = (torch.rand(1, K, device=dev) - 0.5).half()
    t = tune(path=6, **cfg)
    s, _ = timed(lambda: [forward_multi(grp, x, t) for grp in groups])
    out.append(f"{name} {cfg['lanes_n']:2d}/{cfg['waves']:2d}/{cfg['u']}: {s / ng * 1e6:6.2f}")
    del groups
    torch.cuda.empty_cache()
print(os.environ.get("GPTQ_MI355X_LIB", "in-tree").split("/")[-1], " | ".join(out), flush=True)
