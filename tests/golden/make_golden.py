#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ from the REFERENCE itself.

Runs only in the build container (needs /root/reference); the GPU box and the test-suite
only read the produced ``*.npz`` files.  Two kinds of fixture:

1. ``kat_*.npz``   -- the three known-answer vectors that the reference's own tests pin
                     (tests/test_q4.py:29-1056 CUDA_OLD_REFERENCE, :1230-1489 REFERENCE_OLD_HALF,
                     :1491-1750 REFERENCE_OLD_NO_HALF), extracted with ``ast`` (the test module
                     itself cannot be imported here: it needs CUDA extensions + ``parameterized``).
2. ``ref_*.npz``   -- inputs and outputs of the reference ``QuantLinear`` classes
                     (qlinear_cuda_old.py = no act-order, qlinear_cuda.py = act-order), loaded by
                     file path (``import auto_gptq`` fails under transformers 5.x),
                     run on CPU: pack() -> (qweight, qzeros, scales) and forward() -> y, plus the
                     dequantised weight matrix obtained by pushing an identity through forward().

Usage:  python tests/golden/make_golden.py          (re-creates every file, deterministic)
"""
import ast
import importlib.util
import os
import sys

import numpy as np
import torch

REF = os.environ.get("GPTQ_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def load_ref_class(fname):
    path = os.path.join(REF, "auto_gptq/nn_modules/qlinear", fname)
    spec = importlib.util.spec_from_file_location("ref_" + fname[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.QuantLinear


def extract_kats():
    src = open(os.path.join(REF, "tests/test_q4.py")).read()
    tree = ast.parse(src)
    found = {}

    def grab(node):
        # NAME = torch.Tensor([...]).to(torch.float16)   -> list literal
        for sub in ast.walk(node.value):
            if isinstance(sub, ast.List) and len(sub.elts) > 100:
                return [ast.literal_eval(e) for e in sub.elts]
        return None

    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and len(node.targets) == 1:
            t = node.targets[0]
            name = t.id if isinstance(t, ast.Name) else None
            if name in ("CUDA_OLD_REFERENCE", "REFERENCE_OLD_HALF", "REFERENCE_OLD_NO_HALF"):
                vals = grab(node)
                if vals is not None:
                    found[name] = np.asarray(vals, dtype=np.float64)
    assert set(found) == {"CUDA_OLD_REFERENCE", "REFERENCE_OLD_HALF", "REFERENCE_OLD_NO_HALF"}, found.keys()
    assert found["CUDA_OLD_REFERENCE"].shape == (1024,)
    assert found["REFERENCE_OLD_HALF"].shape == (256,)
    assert found["REFERENCE_OLD_NO_HALF"].shape == (256,)
    # the reference stores them as fp16 tensors (.to(torch.float16)) -> round the same way
    np.savez_compressed(os.path.join(HERE, "kat_cuda_old_reference_1024.npz"),
                        y=torch.tensor(found["CUDA_OLD_REFERENCE"]).to(torch.float16).numpy(),
                        k=1024, n=1024, dtype="float16", rtol=3e-5, atol=2e-2,
                        source="tests/test_q4.py:29-1056, recipe :1086-1122")
    np.savez_compressed(os.path.join(HERE, "kat_reference_old_half_256.npz"),
                        y=torch.tensor(found["REFERENCE_OLD_HALF"]).to(torch.float16).numpy(),
                        k=256, n=256, dtype="float16", rtol=1e-3, atol=1e-8,
                        source="tests/test_q4.py:1230-1489, recipe :1752-1802")
    np.savez_compressed(os.path.join(HERE, "kat_reference_old_no_half_256.npz"),
                        y=torch.tensor(found["REFERENCE_OLD_NO_HALF"]).to(torch.float16).numpy(),
                        k=256, n=256, dtype="float32", rtol=1e-3, atol=1e-8,
                        source="tests/test_q4.py:1491-1750, recipe :1752-1802")
    print("kat: 3 vectors written")


def quantizer(W, bits, gs, g_idx, zero_policy):
    """min/max asymmetric quantizer -> scale[N,G], zero[N,G]; zero_policy picks the edge cases
    of SURVEY App. B (#1 wrap at zero == maxq+1, #2 zero == 0 corrupting pack())."""
    N, K = W.shape
    G = int(g_idx.max()) + 1
    maxq = 2 ** bits - 1
    scale = torch.empty(N, G, dtype=W.dtype)
    zero = torch.empty(N, G, dtype=W.dtype)
    for g in range(G):
        cols = W[:, torch.from_numpy(g_idx == g)]
        lo = torch.clamp(cols.min(1).values, max=0)
        hi = torch.clamp(cols.max(1).values, min=0)
        s = (hi - lo) / maxq
        s[s == 0] = 1
        scale[:, g] = s
        zero[:, g] = torch.clamp(torch.round(-lo / s), 1, maxq)
    if zero_policy == "max_plus_one":        # stored field = maxq -> wrap/nowrap fork
        zero[:, ::2] = maxq + 1
    elif zero_policy == "has_zero":          # zero-1 = -1 -> 0xFFFFFFFF OR-ed over the word
        zero[::3, 0] = 0
    return scale, zero


def ref_case(name, QL, *, bits, gs, K, N, act_order, dtype, M, bias, zero_policy="normal", seed=0,
             qparams_dtype=torch.float32):
    """qparams_dtype: dtype of the (scale, zero) handed to pack(); GPTQ hands over fp32
    (quantizer runs on W.float()), so that is the default; fp16 exercises the other promotion."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    W = (torch.randn(N, K) * 0.02).to(torch.float32)
    g_idx = (np.arange(K) // gs).astype(np.int32)
    if act_order:
        g_idx = g_idx[np.random.permutation(K)]
    scale, zero = quantizer(W, bits, gs, g_idx, zero_policy)
    lin = torch.nn.Linear(K, N, bias=bias)
    lin.weight.data = W.clone().to(dtype)
    if bias:
        lin.bias.data = (torch.randn(N) * 0.1).to(dtype)
    q = QL(bits, gs, K, N, bias, weight_dtype=dtype)
    q.pack(lin, scale.clone().to(qparams_dtype), zero.clone().to(qparams_dtype), torch.from_numpy(g_idx.copy()))
    x = (torch.rand(M, K) - 0.5).to(dtype)
    with torch.no_grad():
        y = q(x)
        Wdq = q(torch.eye(K, dtype=dtype))           # row k of the dequantised matrix (+bias)
        if bias:
            Wdq = Wdq - q.bias
    def npf(t):
        t = t.detach()
        return t.float().numpy() if t.dtype == torch.bfloat16 else t.numpy()
    np.savez_compressed(
        os.path.join(HERE, f"ref_{name}.npz"),
        W=W.numpy(), scale=scale.numpy(), zero=zero.numpy(), g_idx=g_idx,
        lin_bias=(npf(lin.bias.data) if bias else np.zeros(0, np.float32)),
        qweight=q.qweight.numpy(), qzeros=q.qzeros.numpy(), scales=npf(q.scales),
        bias=(npf(q.bias) if bias else np.zeros(0, np.float32)),
        x=npf(x), y=npf(y), Wdq=npf(Wdq),
        bits=bits, group_size=gs, K=K, N=N, M=M, act_order=int(act_order),
        dtype=str(dtype).replace("torch.", ""), qparams_dtype=str(qparams_dtype).replace("torch.", ""), ref_class=QL.QUANT_TYPE, zero_policy=zero_policy,
    )
    print(f"ref_{name}: qweight {tuple(q.qweight.shape)} qzeros {tuple(q.qzeros.shape)} y {tuple(y.shape)}")


def main():
    extract_kats()
    old = load_ref_class("qlinear_cuda_old.py")
    new = load_ref_class("qlinear_cuda.py")
    f16, f32, bf16 = torch.float16, torch.float32, torch.bfloat16
    # -- no act-order (cuda_old class) ------------------------------------------------------
    for bits in (2, 3, 4, 8):
        ref_case(f"old_b{bits}_g32_f32", old, bits=bits, gs=32, K=128, N=96, act_order=False, dtype=f32, M=3, bias=False)
        ref_case(f"old_b{bits}_g128_f16", old, bits=bits, gs=128, K=256, N=64, act_order=False, dtype=f16, M=2, bias=True)
    ref_case("old_b4_gfull_f16", old, bits=4, gs=256, K=256, N=64, act_order=False, dtype=f16, M=1, bias=False)
    ref_case("old_b4_g64_bf16", old, bits=4, gs=64, K=128, N=64, act_order=False, dtype=bf16, M=2, bias=False)
    for bits in (2, 3, 4, 8):
        ref_case(f"old_b{bits}_wrap", old, bits=bits, gs=32, K=64, N=64, act_order=False, dtype=f32, M=2, bias=False,
                 zero_policy="max_plus_one")
    ref_case("old_b4_zero0", old, bits=4, gs=32, K=64, N=64, act_order=False, dtype=f32, M=2, bias=False,
             zero_policy="has_zero")
    # -- act-order (cuda class) -------------------------------------------------------------
    for bits in (2, 3, 4, 8):
        ref_case(f"act_b{bits}_g32_f32", new, bits=bits, gs=32, K=128, N=96, act_order=True, dtype=f32, M=3, bias=False)
        ref_case(f"act_b{bits}_g128_f16", new, bits=bits, gs=128, K=256, N=64, act_order=True, dtype=f16, M=2, bias=True)
        ref_case(f"act_b{bits}_wrap", new, bits=bits, gs=32, K=64, N=64, act_order=True, dtype=f32, M=2, bias=False,
                 zero_policy="max_plus_one")
    ref_case("act_b4_g64_bf16", new, bits=4, gs=64, K=128, N=64, act_order=True, dtype=bf16, M=2, bias=False)
    ref_case("old_b4_g128_f16_qp16", old, bits=4, gs=128, K=256, N=64, act_order=False, dtype=f16, M=2, bias=False,
             qparams_dtype=f16)
    ref_case("act_b3_g32_f16_qp16", new, bits=3, gs=32, K=128, N=64, act_order=True, dtype=f16, M=2, bias=False,
             qparams_dtype=f16)
    ref_case("seq_b4_g128_f16_cuda", new, bits=4, gs=128, K=256, N=64, act_order=False, dtype=f16, M=2, bias=False)


if __name__ == "__main__":
    sys.exit(main())
