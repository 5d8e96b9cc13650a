#!/usr/bin/env python3
"""3- / 8-bit fp16 layers, plain and act-order (re-sequenced side copy), M = 1..64 (+ 2048): default plan against the forced
matrix-core GEMV (tuning.path = 5) and the forced MFMA GEMM (tuning.path = 3) -- the crossovers in want_gemm (csrc/capi.hip).
Usage (GPU box): python tools/nonq4_paths.py [--gs 32]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from autogptq_amd import _lib
from tools.gemv_sweep import run

SHAPES = ((4096, 4096), (4096, 11008), (11008, 4096))


def tuning(path):
    t = _lib.GptqTuning()
    t.path = path
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gs", type=int, default=32)
    ap.add_argument("--ms", default="1,2,4,8,16,64,2048")
    ap.add_argument("--plain-only", action="store_true", help="skip the act-order layers")
    args = ap.parse_args()
    ms = [int(v) for v in args.ms.split(",")]
    dev = torch.device("cuda:0")
    for bits in (3, 8):
        for act in ((False,) if args.plain_only else (False, True)):
            for K, N in SHAPES:
                per = K * N * bits // 8
                nl = max(4, min(32, (320 << 20) // per))
                layers = [make_layer(K, N, dev, bits=bits, gs=args.gs, act_order=act, seed=i) for i in range(nl)]
                cells = []
                for M in ms:
                    x = (torch.rand(M, K, device=dev) - 0.5).half()
                    use = layers if M <= 64 else layers[:4]
                    plan = _lib.describe_plan(layers[0]._layer, M)
                    t = {}
                    for name, tn in (("auto", None), ("gemv", tuning(5)), ("gemm", tuning(3))):
                        if (name == "gemv" and M > 16) or (name == "gemm" and M < 2):
                            continue
                        try:
                            t[name] = run(use, x, tn, reps=3) * 1e6
                        except Exception as e:
                            t[name] = float("nan")
                    cells.append(f"{M}:" + "/".join(f"{t[k]:.1f}" for k in ("auto", "gemv", "gemm") if k in t) + f"[{plan['kernel'][:6]}]")
                print(f"int{bits} g{args.gs} act={int(act)} {K}x{N} (us auto/gemv/gemm):  " + "  ".join(cells), flush=True)
                del layers
                torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
