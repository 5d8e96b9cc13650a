#!/usr/bin/env python3
"""3- / 8-bit fp16 decode: packed magic-number field decode (default) against the field-by-field form (tuning.reserved[1] = 1),
interleaved, on rotating HBM-cold weights inside a hipGraph; then waves / K-split around the default plan for the 3-bit layers.
Usage (GPU box): python tools/magic_ab.py [--gs 32] [--rounds 3]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer, algorithmic_bytes
from autogptq_amd import _lib
from tools.gemv_sweep import run

SHAPES = ((4096, 4096), (4096, 11008), (11008, 4096))


def tuning(**kw):
    t = _lib.GptqTuning()
    for k, v in kw.items():
        if k == "r1":
            t.reserved[1] = v
        else:
            setattr(t, k, v)
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gs", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--bits", default="3,8")
    ap.add_argument("--sweep-only", action="store_true", help="skip the magic / field-by-field A/B, only the waves x K-split sweep")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for bits in [int(b) for b in args.bits.split(',')]:
        for K, N in SHAPES:
            per = K * N * bits // 8
            nl = max(4, min(48, (512 << 20) // per))
            layers = [make_layer(K, N, dev, bits=bits, gs=args.gs, seed=i) for i in range(nl)]
            for M in (() if args.sweep_only else (1, 4)):
                x = (torch.rand(M, K, device=dev) - 0.5).half()
                ab = algorithmic_bytes(K, N, M, bits=bits, gs=args.gs)
                with torch.no_grad():
                    y_new = layers[0](x, tuning=tuning())
                    y_old = layers[0](x, tuning=tuning(r1=1))
                torch.cuda.synchronize()
                diff = float((y_new.float() - y_old.float()).abs().max() / y_old.float().abs().max())
                plan = _lib.describe_plan(layers[0]._layer, M)
                best = {"magic": 1e9, "bfe": 1e9}
                for _ in range(args.rounds):
                    best["magic"] = min(best["magic"], run(layers, x, tuning()))
                    best["bfe"] = min(best["bfe"], run(layers, x, tuning(r1=1)))
                print(f"int{bits} g{args.gs} {K}x{N} M={M}: magic {best['magic']*1e6:7.2f} us ({ab/best['magic']/1e9:7.1f} GB/s)  "
                      f"field-by-field {best['bfe']*1e6:7.2f} us ({ab/best['bfe']/1e9:7.1f} GB/s)  rel diff {diff:.2e}  "
                      f"{plan.get('kernel')} waves={plan.get('waves')} u={plan.get('u')} deq={plan.get('deq')}", flush=True)
            if not args.no_sweep:
                x = (torch.rand(1, K, device=dev) - 0.5).half()
                res = []
                for waves in (4, 8, 16):
                    for ks in (1, 2, 4):
                        try:
                            res.append((run(layers, x, tuning(path=5, waves=waves, ksplit=ks)), waves, ks))
                        except Exception as e:
                            print("  fail", waves, ks, repr(e)[:100])
                res.sort()
                print("   sweep (M=1, us/waves/ksplit): " + "  ".join(f"{s*1e6:.2f}/{w}/{k}" for s, w, k in res), flush=True)
            del layers
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
