#!/usr/bin/env python3
"""ONE prefill layer shape, a few layer calls (x permute of act-order layers + GEMM) on rotating layers in a hipGraph: the command the per-shape rocprofv3
passes of tools/session_r05_prof.sh wrap -- the stream-K prefill kernel launches one workgroup per CU whatever the shape, so its dispatches can only be
labelled with (K, N, M) by running one shape per database.  Usage: python tools/prefill_one.py --k 4096 --n 11008 --m 2048 [--act] [--dtype f16] [--bits 4] [--gs 128]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_layer
from tools.gemv_sweep import run
from autogptq_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, required=True)
ap.add_argument("--n", type=int, required=True)
ap.add_argument("--m", type=int, required=True)
ap.add_argument("--act", action="store_true")
ap.add_argument("--dtype", default="f16")
ap.add_argument("--bits", type=int, default=4)
ap.add_argument("--gs", type=int, default=128)
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
ls = [make_layer(a.k, a.n, dev, bits=a.bits, gs=a.gs, dtype=dt, seed=i, act_order=a.act) for i in range(a.layers)]
x = (torch.rand(a.m, a.k, device=dev) - 0.5).to(dt)
s = min(run(ls, x, None, reps=a.reps) for _ in range(2))
d = _lib.describe_plan(ls[0]._layer, a.m)
print(f"int{a.bits} g{a.gs} {a.k}x{a.n} M={a.m} act={int(a.act)} {a.dtype}: {s * 1e6:.1f} us per layer call, {2 * a.m * a.k * a.n / s / 1e12:.0f} TFLOP/s, plan {d}")
