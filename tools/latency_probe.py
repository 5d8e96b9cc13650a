#!/usr/bin/env python3
"""Is a GEMM K-step bound by the latency of its (HBM-cold) weight words?  Same launches twice: over rotating layers (> 256 MiB of
distinct weights: every weight byte comes from HBM) and over ONE layer replayed (weights hot in the Infinity Cache / L2).  If the step
chain is latency-bound the hot run is much faster although the kernel does identical work.
Usage (GPU box): python tools/latency_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from autogptq_amd import _lib
from tools.gemv_sweep import run


def tuning(**kw):
    t = _lib.GptqTuning()
    for k, v in kw.items():
        setattr(t, k, v)
    return t


def main():
    dev = torch.device("cuda:0")
    cases = [(4096, 4096, 2048, None, "prefill, default plan"), (4096, 11008, 2048, None, "prefill, default plan"),
             (4096, 4096, 4096, None, "north_star M = 4096"),
             (4096, 4096, 128, dict(path=3, ksplit=1), "mid-M, forced ksplit = 1 (32 steps per K group)"),
             (4096, 4096, 128, None, "mid-M, default plan"), (4096, 11008, 128, None, "mid-M, default plan"),
             (4096, 4096, 1, None, "decode, default plan"), (11008, 4096, 1, None, "decode, default plan")]
    for K, N, M, tn, what in cases:
        per = K * N // 2
        nl = max(4, min(48, (384 << 20) // per))
        layers = [make_layer(K, N, dev, seed=i) for i in range(nl)]
        x = (torch.rand(M, K, device=dev) - 0.5).half()
        t = tuning(**tn) if tn else None
        cold = min(run(layers, x, t, reps=3) for _ in range(2))
        hot = min(run([layers[0]] * 16, x, t, reps=3) for _ in range(2))
        plan = _lib.describe_plan(layers[0]._layer, M, t)
        steps = (K // 64) // max(1, plan.get("ksplit", 1)) // (2 if plan.get("kg", 1) == 2 else 1) if plan.get("path") == "gemm" else 0
        print(f"{K}x{N} M={M:5d} {what:48s} cold {cold*1e6:8.2f} us  hot {hot*1e6:8.2f} us  ratio {cold/hot:4.2f}  "
              f"{plan.get('kernel')} ksplit={plan.get('ksplit')} kg={plan.get('kg')}"
              + (f"  K-steps per group {steps}: {cold*1e6/steps:.2f} -> {hot*1e6/steps:.2f} us per step" if steps else ""), flush=True)
        del layers
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
