#!/usr/bin/env python3
"""Balanced tail of the big-tile prefill kernel (tuning.reserved[3] = 40 the planner's rule / 41 off): interleaved A/B on rotating layers inside a hipGraph (HIP
events, min over rounds), every output of the two forms compared, repeat launches compared bit for bit, header words checked.
Usage: python tools/tail_ab.py [--cases 4096x11008x2048a,...] [--rounds 3]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from autogptq_amd import _lib
from autogptq_amd import qlinear_mi355x as QM
from tools.gemv_sweep import run


def tun(v):
    t = _lib.GptqTuning()
    t.path = 3
    t.reserved[3] = v
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="4096x11008x768a,4096x4096x2176,4096x4096x2176a,11008x4096x2176a,4096x11008x1024,4096x11008x1536a,4096x11008x1664a,4096x11008x2304,4096x4096x4224,4096x4096x2304ab,4096x11008x2048a")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--gs", type=int, default=128)
    ap.add_argument("--on", type=int, default=40, help="40 = the planner's rule, 42 = the rule without its tile limit")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for case in args.cases.split(","):
        bf = case.endswith("b")
        c = case.rstrip("b")
        act = c.endswith("a")
        K, N, M = map(int, c.rstrip("a").split("x"))
        dt = torch.bfloat16 if bf else torch.float16
        n = 6 if K * N <= 64 << 20 else 3
        layers = [make_layer(K, N, dev, bits=args.bits, gs=args.gs, act_order=act, dtype=dt, seed=i) for i in range(n)]
        x = (torch.rand(M, K, device=dev) - 0.5).to(dt)
        t_on, t_off = tun(args.on), tun(41)
        plan = _lib.describe_plan(layers[0]._layer, M, t_on)
        best = {40: 1e9, 41: 1e9}
        for _ in range(args.rounds):
            for v, t in ((41, t_off), (40, t_on)):
                best[v] = min(best[v], run(layers, x, t, reps=4))
        with torch.no_grad():
            y_off = layers[0](x, tuning=t_off)
            y_on = [layers[0](x, tuning=t_on) for _ in range(3)]
        torch.cuda.synchronize()
        same = all(torch.equal(y_on[0], y) for y in y_on[1:])
        d = (y_on[0].float() - y_off.float()).abs()
        scale = float(y_off.float().abs().max())
        hdr_ok, err = True, 0
        for ent in QM._WORKSPACE.values():
            h = ent[0][:_lib.WS_HEADER_BYTES].view(torch.int32)
            hdr_ok &= int(h[:8192].abs().max().item()) == 0
            err |= int(h[_lib.WS_HEADER_BYTES // 4 - 14].item())
        fl = 2 * M * K * N
        print(f"int{args.bits} g{args.gs} {K}x{N} M={M} act={int(act)} {'bf16' if bf else 'f16'} tiles={plan['tiles']} tail={plan['tail']}x{plan['tail_slices']}: whole {best[41] * 1e6:7.1f} us {fl / best[41] / 1e12:6.0f} TF | "
              f"balanced {best[40] * 1e6:7.1f} us {fl / best[40] / 1e12:6.0f} TF ({best[41] / best[40]:.3f}x)  max|diff| {float(d.max()):.3g} of {scale:.3g}, "
              f"differing {int((d > 0).sum())}/{d.numel()}, repeat-identical={same} header_zero={hdr_ok} err={err}", flush=True)
        del layers, x, y_off, y_on
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
