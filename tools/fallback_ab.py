#!/usr/bin/env python3
"""Round 6: the checkpoint-layout fallbacks whose code changed when the spilling instantiations were retired (gemv_generic_kernel's per-k form as a rolled loop,
the streamed GEMVs' accumulators as one vector per row, skinny 3-bit / stream64 forms): us per layer call of the DEFAULT plan on rotating layers in a hipGraph.
Run once per library (GPTQ_MI355X_LIB=tools/libgptq_r6pre.so for the tree before the change): tools/session_r06_sprawl.sh interleaves the two.
Prints the first layer's output checksum too: the two libraries must agree bit for bit (same fma order per accumulator)."""
import os, sys, hashlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import autogptq_amd
from autogptq_amd import _lib
from tools.gemv_sweep import run

dev = torch.device("cuda:0")
DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}


def layer(K, N, bits, gs, dt, seed, act, tiled):
    g = torch.Generator(device=dev).manual_seed(seed)
    q = autogptq_amd.QuantLinear(bits, gs, K, N, False, weight_dtype=dt)
    G = -(-K // gs)
    q.qweight = torch.randint(-2**31, 2**31 - 1, (K // 32 * bits, N), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    q.qzeros = torch.randint(-2**31, 2**31 - 1, (G, N // 32 * bits), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    q.scales = (0.002 * (1 + 0.1 * torch.rand(G, N, device=dev, generator=g))).to(dt)
    gi = torch.arange(K, device=dev, dtype=torch.int32) // gs
    if act:
        gi = gi[torch.randperm(K, device=dev, generator=g)]
    q.g_idx = gi.contiguous()
    q = q.to(dev)
    q.post_init(tiled=tiled)
    return q


# (bits, group_size, dtype, K, N, M, act-order, decode copy)
CASES = [
    (3, 16, "f32", 4096, 4096, 1, False, True),
    (3, 16, "f16", 4096, 4096, 1, False, True),
    (3, 16, "f16", 4096, 4096, 4, False, True),
    (2, 8, "f32", 4096, 4096, 4, False, True),
    (4, 12, "f16", 4032, 4096, 2, False, True),
    (3, 128, "f32", 4096, 4096, 2, False, True),
    (4, 128, "f32", 4096, 11008, 1, False, True),
    (4, 128, "f32", 4096, 11008, 4, False, True),
    (4, 96, "f16", 4032, 4096, 1, False, True),
    (2, 64, "f16", 4096, 11008, 1, False, True),
    (2, 64, "f16", 4096, 11008, 2, False, True),
    (2, 64, "f16", 4096, 11008, 4, False, True),
    (4, 128, "f16", 8192, 8192, 2, False, False),
    (4, 128, "f16", 8192, 8192, 4, False, False),
    (4, 128, "f16", 4096, 4096, 1, False, False),
    (4, 128, "f16", 4096, 11008, 1, True, False),
    (4, 128, "f16", 4096, 4096, 8, False, False),
    (3, 64, "f16", 1024, 1024, 64, False, True),
    (4, 128, "f16", 4096, 11008, 64, False, False),
]

print(f"library: {_lib.LIB_PATH}")
for bits, gs, dts, K, N, M, act, tiled in CASES:
    dt = DT[dts]
    nl = max(4, min(48, (640 << 20) // (K * N * bits // 8)))
    ls = [layer(K, N, bits, gs, dt, i, act, tiled) for i in range(nl)]
    x = (torch.rand(M, K, device=dev, generator=torch.Generator(device=dev).manual_seed(7)) - 0.5).to(dt)
    d = _lib.describe_plan(ls[0]._layer, M)
    with torch.no_grad():
        y = ls[0](x)
    torch.cuda.synchronize()
    h = hashlib.sha256(y.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:12]
    t = min(run(ls, x, None, reps=5) for _ in range(3))
    print(f"int{bits} g{gs:<3d} {dts:4s} {K}x{N} M={M:3d} act={int(act)} copy={int(tiled)} | {d.get('kernel'):12s} ln={d.get('ln', '-')} mt={d.get('mt')} u={d.get('u')} waves={d.get('waves')} ks={d.get('ksplit')} | {t * 1e6:8.2f} us | out {h}", flush=True)
    del ls, x
