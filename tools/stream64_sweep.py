#!/usr/bin/env python3
"""Batched decode (4 < M <= 64): gemm_stream64_kernel launch geometries against the kernels it replaces, rotating HBM-cold layers
in a hipGraph.  Usage: python tools/stream64_sweep.py [--ms 8,16,32,64] [--shapes 4096x4096,...] [--dtype f16] [--act] [--quick]"""
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
    ap.add_argument("--ms", default="5,8,16,32,64")
    ap.add_argument("--shapes", default="4096x4096,4096x11008,11008x4096")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--act", action="store_true")
    ap.add_argument("--quick", action="store_true", help="default plans only")
    ap.add_argument("--ks", default="", help="K-slice counts to sweep (default 1,2,3,4,6,8)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    for shp in a.shapes.split(","):
        K, N = map(int, shp.split("x"))
        nl = max(4, min(32, (400 << 20) // (K * N // 2)))
        ls = [make_layer(K, N, dev, act_order=a.act, dtype=dt, seed=i) for i in range(nl)]
        for M in map(int, a.ms.split(",")):
            x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
            ab = algorithmic_bytes(K, N, M, act_order=a.act)
            out = []
            auto = run(ls, x, None)
            out.append(f"auto[{_lib.describe_plan(ls[0]._layer, M).get('kernel')}]={auto * 1e6:.2f}")
            olds = [("strip16", tun(path=3, reserved={2: 3})), ("skinny", tun(path=3, reserved={2: 1})), ("tiled", tun(path=3, reserved={2: 2}))]
            if M <= 8:
                olds.append(("gemv", tun(path=5)))
            for name, t in olds:
                try:
                    out.append(f"{name}={run(ls, x, t) * 1e6:.2f}")
                except Exception as e:
                    out.append(f"{name}=n/a")
            res = []
            if not a.quick:
                rt = 1 if M <= 16 else (2 if M <= 32 else 4)
                for waves in ((8, 16) if rt == 1 else (4, 8)):
                    for u in ((2, 4, 8) if rt == 1 else ((2, 4) if rt == 2 else (1, 2, 4))):
                        for ks in ((1, 2, 3, 4, 6, 8) if not a.ks else tuple(map(int, a.ks.split(",")))):
                            t = tun(path=3, waves=waves, ksplit=ks, reserved={0: u, 2: 4})
                            tag = f"w{waves}u{u}k{ks}"
                            try:
                                res.append((run(ls, x, t), tag))
                            except Exception as e:
                                res.append((9.9, tag + ":FAIL"))
                res.sort()
            best = " ".join(f"{n}={s * 1e6:.2f}" for s, n in res[:10])
            print(f"{K}x{N} M={M:2d} ({ab / 1e6:.1f} MB; {ab / auto / 1e9:.0f} GB/s auto): " + " ".join(out) + " | " + best, flush=True)
        del ls


if __name__ == "__main__":
    main()
