#!/usr/bin/env python3
"""Golden fixtures for the Marlin checkpoint format, produced by the REFERENCE's own ``QuantLinear.pack``
(auto_gptq/nn_modules/qlinear/qlinear_marlin.py:133-176), loaded by file path.  Its constructor refuses to run without an
sm80 CUDA device (:90-94), so the module object is created without calling it (``nn.Module.__init__`` + the three attributes
and two buffers ``pack`` touches); ``pack`` itself is plain CPU tensor/numpy code.

Stored: the fake-quantised fp16 weight [N, K] and scales handed to ``pack`` and the resulting Marlin tensors ``B int32
[K/16, 2N]`` and ``s fp16 [G, N]``.   Usage:  python tests/golden/make_golden_marlin.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = os.environ.get("GPTQ_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_marlin_class():
    path = os.path.join(REF, "auto_gptq/nn_modules/qlinear/qlinear_marlin.py")
    spec = importlib.util.spec_from_file_location("ref_qlinear_marlin", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.QuantLinear


def case(QL, name, K, N, gs, seed, bias):
    g = torch.Generator().manual_seed(seed)
    G = K // gs
    W = torch.randn(N, K, generator=g) * 0.05
    s = (W.reshape(N, G, gs).abs().amax(dim=2) / 7 + 1e-4).half()                 # [N, G], symmetric int4
    ints = torch.clamp(torch.round(W / s.float().repeat_interleave(gs, 1)), -8, 7)
    Wq = (ints * s.float().repeat_interleave(gs, 1)).half()                       # fake-quantised weight
    lin = torch.nn.Linear(K, N, bias=bias).half()
    lin.weight.data = Wq.clone()
    if bias:
        lin.bias.data = (torch.randn(N, generator=g) * 0.1).half()
    q = QL.__new__(QL)
    torch.nn.Module.__init__(q)
    q.infeatures, q.outfeatures, q.group_size = K, N, gs
    q.register_buffer("B", torch.empty((K // 16, N * 16 // 8), dtype=torch.int))
    q.register_buffer("s", torch.empty((K // gs, N), dtype=torch.half))
    q.bias = None
    q.pack(lin, s.clone())
    np.savez_compressed(os.path.join(HERE, f"marlin_{name}.npz"), Wq=Wq.numpy(), scales=s.numpy(), ints=(ints + 8).to(torch.uint8).numpy(),
                        B=q.B.numpy(), s=q.s.numpy(), bias=(q.bias.detach().numpy() if q.bias is not None else np.zeros(0, np.float16)),
                        K=K, N=N, group_size=gs)
    print(f"marlin_{name}: B {tuple(q.B.shape)} s {tuple(q.s.shape)}")


def main():
    QL = load_marlin_class()
    case(QL, "k128_n256_g128", 128, 256, 128, 0, False)          # group_size == K: the single-group scale permutation
    case(QL, "k256_n256_g128", 256, 256, 128, 1, True)
    case(QL, "k512_n512_g128", 512, 512, 128, 2, False)


if __name__ == "__main__":
    sys.exit(main())
