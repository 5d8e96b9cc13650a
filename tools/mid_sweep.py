#!/usr/bin/env python3
"""17 .. 128 rows of x: gemm_mid_kernel (tuning.path = 3, reserved[2] = 5) against the planner's default and the older kernels, rotating HBM-cold
layers in a hipGraph, plus a correctness check of every forced configuration against the fp64 product with the layer's own dequantised W.
Usage: python tools/mid_sweep.py [--ms 33,48,64,96,128] [--shapes 4096x4096,4096x11008,11008x4096] [--dtype f16] [--act] [--quick]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer, algorithmic_bytes
from tools.gemv_sweep import run
from autogptq_amd import _lib


def tun(**kw):
    t = _lib.GptqTuning()
    r = kw.pop("reserved", {})
    for k, v in kw.items():
        setattr(t, k, v)
    for i, v in r.items():
        t.reserved[i] = v
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="17,32,33,48,64,96,128")
    ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--act", action="store_true")
    ap.add_argument("--quick", action="store_true", help="default geometry of the mid kernel only")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    bad = 0
    for shp in a.shapes.split(","):
        K, N = map(int, shp.split("x"))
        nl = max(4, min(32, (400 << 20) // (K * N // 2)))
        ls = [make_layer(K, N, dev, act_order=a.act, dtype=dt, seed=i) for i in range(nl)]
        W64 = ls[0].dequantize().double()
        for M in map(int, a.ms.split(",")):
            x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
            ref = x.double() @ W64
            scale = float(ref.abs().max())
            ab = algorithmic_bytes(K, N, M, act_order=a.act)
            out = []
            auto = run(ls, x, None)
            out.append(f"auto[{_lib.describe_plan(ls[0]._layer, M).get('kernel')}]={auto * 1e6:.2f}")
            for name, t in (("s64", tun(path=3, reserved={2: 4})), ("skinny", tun(path=3, reserved={2: 1})), ("tiled", tun(path=3, reserved={2: 2}))):
                try:
                    out.append(f"{name}={run(ls, x, t) * 1e6:.2f}")
                except Exception as e:
                    out.append(f"{name}=n/a")
            res = []
            cfgs = [(0, 0, 1, 0, 0)] if a.quick else [(2, ks, 1, 0, rb) for rb in (1, 2, 4, 8) for ks in (0, 1, 2, 4)]
            for st, ks, cw, fl, rb in cfgs:
                if cw == 2 and (M > 64 or N % 128):
                    continue
                if rb > (M + 15) // 16:
                    continue
                t = tun(path=3, ksplit=ks, lanes_n=rb, reserved={0: st, 1: fl, 2: 5, 3: cw})
                tag = f"rb{rb}k{ks}"
                try:
                    with torch.no_grad():
                        y = ls[0](x, tuning=t)
                    torch.cuda.synchronize()
                    err = float((y.double() - ref).abs().max())
                    tol = (2e-3 if dt == torch.float16 else 1.6e-2) * scale
                    ok = err <= tol and bool(torch.isfinite(y).all())
                    if not ok:
                        bad += 1
                        tag += f":WRONG(err={err:.3g},scale={scale:.3g})"
                    res.append((run(ls, x, t), tag))
                except Exception as e:
                    res.append((9.9, tag + ":FAIL " + str(e)[:60]))
            res.sort()
            best = " ".join(f"{n}={s * 1e6:.2f}" for s, n in res[:10])
            print(f"{K}x{N} M={M:3d} ({ab / 1e6:.1f} MB): " + " ".join(out) + " | mid: " + best, flush=True)
        del ls
    print("WRONG RESULTS:", bad)


if __name__ == "__main__":
    main()
