#!/usr/bin/env python3
"""Balanced tail of gemm_wide_kernel (tuning.reserved[3] = 48, lab knob): the tiles past the last full round of 256 as two K halves each, combined inside the
launch.  One GPU call: (1) parity -- a small ragged layer against x (fp64) @ W_oracle (fp64), plain and act-order, with bias, bit-reproducible; the BASELINE
config-3 layer (4096 x 11008, M = 2048, desc_act) against the default plan's output; (2) time -- default plan (128 x 256 tiles) against wide tiles without
and with the tail, interleaved rounds on rotating layers in a hipGraph, settled clocks.  Usage: python tools/wide_tail_ab.py [--quick]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import QuantLinear
from oracle import gptq_oracle as O

dev = torch.device("cuda:0")


def tune(v):
    if v is None:
        return None
    t = _lib.GptqTuning()
    t.path, t.reserved[3], t.ksplit = 3, v, 1
    return t


ok = True
# ---- (1a) small layers, every output against the fp64 product (260, 258 and 325 wide tiles: remainders 4, 2 -- with a ragged last row and column tile -- and 69)
for (K, N, M, act, gs) in ((256, 1024, 128 * 130, False, 128), (512, 1056, 128 * 86 - 77, True, 128), (1024, 2560, 128 * 65, False, 256)):
    Lq = O.random_quant_layer(K, N, 4, gs, act_order=act, seed=K + N, bias=True)
    q = QuantLinear(4, gs, K, N, True)
    q.qweight, q.qzeros, q.scales, q.g_idx, q.bias = Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], Lq["bias"]
    q = q.to(dev)
    q.post_init()
    plan = _lib.describe_plan(q._layer, M, tune(48))
    W = O.dequantize(Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], 4, O.ZERO_NOWRAP if act else O.ZERO_WRAP).to(dev)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half().to(dev)
    for r in range(0, M, 97):                                  # one-hot rows through whole and tail tiles
        x[r].zero_()
        x[r, (r * 7 + 3) % K] = 1.0
    with torch.no_grad():
        y, y2 = q(x, tuning=tune(48)), q(x, tuning=tune(48))
    ref = x.double() @ W.double() + Lq["bias"].to(dev).double()
    scale = float(ref.abs().max())
    bad = (y.double() - ref).abs() > 1e-3 * scale + 1e-3 * ref.abs()
    good = torch.equal(y, y2) and not bool(bad.any()) and int(plan.get("tail", 0)) > 0
    ok &= good
    print(f"parity {K}x{N} M={M} act={act}: plan {plan.get('kernel')} tiles {plan.get('tiles')} tail {plan.get('tail')}  reproducible {torch.equal(y, y2)}  out of tolerance {int(bad.sum())}/{bad.numel()}  -> {'ok' if good else 'FAIL'}", flush=True)
    del q, W, x, y, y2, ref

# ---- (1b) + (2) the BASELINE config-3 layer
K, N, M = 4096, 11008, 2048
nl = 4 if "--quick" in sys.argv else 12
ls = [make_layer(K, N, dev, seed=i, act_order=True) for i in range(nl)]
x = (torch.rand(M, K, device=dev) - 0.5).half()
with torch.no_grad():
    y_def = ls[0](x)
    y_tail, y_tail2 = ls[0](x, tuning=tune(48)), ls[0](x, tuning=tune(48))
d = (y_tail.double() - y_def.double()).abs()
scale = float(y_def.double().abs().max())
good = torch.equal(y_tail, y_tail2) and float(d.max()) <= 2e-3 * scale
ok &= good
print(f"parity {K}x{N} M={M} desc_act: plan {_lib.describe_plan(ls[0]._layer, M, tune(48))}\n   reproducible {torch.equal(y_tail, y_tail2)}  max |tail - default| / max |y| = {float(d.max()) / scale:.2e} -> {'ok' if good else 'FAIL'}", flush=True)
if ok:
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        with torch.no_grad():
            ls[0](x)
        torch.cuda.synchronize()
    best = {}
    for _ in range(3):
        for name, v in (("128 x 256 tiles (default)", None), ("wide tiles, two rounds", 45), ("wide tiles + balanced tail", 48)):
            s = run(ls, x, tune(v), reps=3)
            best[name] = min(best.get(name, 1e9), s)
    print(f"{K}x{N} M={M} desc_act, layer call incl. the x permute: " + "   ".join(f"{k}: {v * 1e6:7.1f} us {2 * M * K * N / v / 1e12:6.0f} TF" for k, v in best.items()), flush=True)
print("RESULT", "ok" if ok else "FAIL")
