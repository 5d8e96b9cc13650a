#!/usr/bin/env python3
"""q|k|v and gate|up at 17..128 rows: gptq_forward_multi (one gemm_mid_kernel launch where the planner takes it) against the layers one by one,
rotating HBM-cold weights in a hipGraph.  Usage: python tools/mid_multi_ab.py [--ms 32,64,128]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.stream_sweep import timed
from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import forward_multi


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="17,32,64,96,128")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for gname, K, Ns in (("q|k|v", 4096, (4096, 4096, 4096)), ("gate|up", 4096, (11008, 11008))):
        ng = max(3, (400 << 20) // (K * sum(Ns) // 2))
        groups = [[make_layer(K, n, dev, seed=100 * gi + i) for i, n in enumerate(Ns)] for gi in range(ng)]
        for M in map(int, a.ms.split(",")):
            x = (torch.rand(M, K, device=dev) - 0.5).half()
            one, _ = timed(lambda: [forward_multi(g, x) for g in groups])
            sep, _ = timed(lambda: [[q(x) for q in g] for g in groups])
            frc = []
            for rb in (1, 2, 3, 4, 8):
                if rb > (M + 15) // 16:
                    continue
                t = _lib.GptqTuning(); t.path = 3; t.reserved[2] = 5; t.lanes_n = rb
                try:
                    f, _ = timed(lambda: [forward_multi(g, x, t) for g in groups])
                    frc.append(f"rb{rb}={f / ng * 1e6:.2f}")
                except Exception as e:
                    frc.append(f"rb{rb}=n/a({str(e)[:40]})")
            print(f"{gname:8s} M={M:3d}: forward_multi (default) {one / ng * 1e6:7.2f} us | layer by layer {sep / ng * 1e6:7.2f} us "
                  f"[{_lib.describe_plan(groups[0][0]._layer, M).get('kernel')}] | forced one mid launch: " + " ".join(frc), flush=True)
        del groups
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
