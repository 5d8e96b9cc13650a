"""GPU (-m gpu): randomised DEFAULT plans against the fp64 oracle.

Round 6 cut the library to the kernel instantiations a plan can ask for (tests/test_kernel_resources.py): a plan whose instantiation went missing is a LAUNCH
error that no host-only planner test can see.  This file draws seeded random layer configurations -- every bit width, group sizes that are and are not whole
packing units or powers of two, fp16 / bf16 / fp32, plain / act-order, with and without the decode copy, [gate | up] layers with the fused epilogue, the row
counts on both sides of every planner threshold, small and Llama-sized shapes -- runs the planner's own choice through the C ABI and compares EVERY output with
the oracle's fp64 product (exact dequant: oracle/gptq_oracle.py:forward_f64, the reference's math of qlinear_cuda.py:253-317 without its fp16 roundings).
Tolerances: the file-wide ones of test_gpu_parity.py (fp16 1e-3, bf16 8e-3, fp32 1e-4, relative to the largest output, sqrt(K / 1024) growth).  The kernels
the plans landed on are counted: a run that never reached a family fails, so the fuzz cannot silently shrink."""
import random

import pytest
import torch

from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import QuantLinear, forward_multi
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float16: (1e-3, 1e-3), torch.bfloat16: (8e-3, 8e-3), torch.float32: (1e-4, 1e-4)}
SEEN = {}


def _close(y, y64, dtype, K, what):
    rtol, atol = TOL[dtype]
    scale = max(1.0, float(y64.abs().max()))
    a = atol * scale * max(1.0, (K / 1024) ** 0.5)
    d = (y.double().cpu() - y64).abs()
    bad = d > a + rtol * y64.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} out of tolerance, max abs diff {float(d.max())} (allowed {a:.3g} + {rtol} |y|)"


def _w64(L, bits, mode):
    """The layer's exact weights in fp64 (oracle.forward_f64's dequant, once per layer): scales[g] * (w - z[g])."""
    import numpy as np
    w = O.unpack_weights(L["qweight"], bits).astype(np.int32)
    z = O.unpack_zeros(L["qzeros"], bits, mode)
    g = L["g_idx"].cpu().numpy().astype(np.int64)
    return L["scales"].double()[torch.from_numpy(g)] * torch.from_numpy((w - z[g]).astype(np.float64))


def _module(L, bits, gs, dtype, tiled, epilogue="none", zero_mode="wrap"):
    K, N = L["K"], L["N"]
    q = QuantLinear(bits, gs, K, N, L["bias"] is not None, weight_dtype=dtype, zero_mode=zero_mode, epilogue=epilogue)
    q.qweight, q.qzeros, q.scales, q.g_idx = L["qweight"].clone(), L["qzeros"].clone(), L["scales"].clone(), L["g_idx"].clone().to(torch.int32)
    if L["bias"] is not None:
        q.bias = L["bias"].clone()
    q = q.to(DEV)
    q.post_init(tiled=tiled)
    return q


def _draw(rnd, big):
    bits = rnd.choice((2, 3, 4, 4, 4, 8))
    dtype = rnd.choice((torch.float16, torch.float16, torch.bfloat16, torch.float32))
    if big:
        K, N = rnd.choice(((4096, 4096), (4096, 11008), (11008, 4096), (2048, 8192), (8192, 2048), (4096, 12288), (5120, 5120)))
        gs = rnd.choice((32, 64, 128, 128, K))
    else:
        K = rnd.choice((64, 96, 160, 256, 512, 1024, 1056, 2048, 2112, 4160))
        N = rnd.choice((32, 64, 96, 256, 1024, 1056, 2048))
        kpu = 32 if bits == 3 else 32 // bits
        gs = rnd.choice([g for g in (8, 12, 16, 24, 32, 48, 64, 96, 128, 256, K) if K % g == 0 and (bits != 3 or g % 32 == 0 or rnd.random() < 0.3)] or [K])
        del kpu
    act = rnd.random() < 0.3
    tiled = rnd.random() < 0.75
    epi = "silu_mul" if (N % 64 == 0 and rnd.random() < 0.12) else "none"
    Ms = rnd.sample((1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 33, 64, 65, 100, 128, 129, 200, 256, 300, 512, 600, 768, 1024) if not big else
                    (1, 2, 3, 4, 5, 8, 16, 33, 64, 100, 128, 130, 256, 384, 512, 768), 4 if big else 5)
    return bits, dtype, K, N, gs, act, tiled, epi, sorted(Ms)


@pytest.mark.parametrize("seed,big", [(s, False) for s in range(200)] + [(1000 + s, True) for s in range(48)])
def test_random_default_plans_against_the_oracle(seed, big):
    rnd = random.Random(seed)
    bits, dtype, K, N, gs, act, tiled, epi, Ms = _draw(rnd, big)
    try:
        L = O.random_quant_layer(K, N, bits, gs, act_order=act, dtype=dtype, seed=seed, bias=rnd.random() < 0.5)
    except Exception as e:                                      # a configuration the checkpoint layout itself cannot hold
        pytest.skip(f"oracle cannot build it: {e}")
    L.setdefault("act_order", act)
    # the zero-point convention is spelled out ("auto" = the reference class's own: cuda_old wraps except in its 3-bit branch, the act-order class never does --
    # pinned in test_gpu_parity.py); both conventions on every packing here
    zm = "nowrap" if act else rnd.choice(("wrap", "nowrap"))
    try:
        q = _module(L, bits, gs, dtype, tiled, epi, zm)
    except (_lib.GptqError, ValueError) as e:
        pytest.skip(f"refused at post_init (documented limits): {e}")
    mode = O.ZERO_NOWRAP if zm == "nowrap" else O.ZERO_WRAP
    W64 = _w64(L, bits, mode)
    for M in Ms:
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(seed * 31 + M)) - 0.5).to(dtype)
        y64 = x.double() @ W64 + (L["bias"].double() if L["bias"] is not None else 0.0)
        if epi == "silu_mul":
            g_, u_ = y64[:, :N // 2], y64[:, N // 2:]
            y64 = g_ / (1.0 + torch.exp(-g_)) * u_
        d = _lib.describe_plan(q._layer, M)
        with torch.no_grad():
            y, y2 = q(x.to(DEV)), q(x.to(DEV))
        assert torch.equal(y, y2), f"not reproducible: {d}"
        SEEN[d.get("kernel")] = SEEN.get(d.get("kernel"), 0) + 1
        _close(y, y64, dtype, K, f"int{bits} g{gs} {K}x{N} M={M} {dtype} act={act} zero={zm} copy={tiled} epi={epi} plan={d}")


@pytest.mark.parametrize("seed", range(40))
def test_random_multi_layer_launches_against_the_oracle(seed):
    """gptq_forward_multi (q|k|v / gate|up callers): 2..4 layers of one packing that read the same x, default plans."""
    rnd = random.Random(500 + seed)
    bits = rnd.choice((2, 3, 4, 4, 8))
    dtype = rnd.choice((torch.float16, torch.bfloat16))
    K = rnd.choice((512, 1024, 2048, 4096))
    gs = rnd.choice((32, 64, 128))
    widths = [rnd.choice((64, 256, 512, 1024, 4096)) for _ in range(rnd.choice((2, 3, 4)))]
    tiled = rnd.random() < 0.8
    Ls = [O.random_quant_layer(K, n, bits, gs, dtype=dtype, seed=900 + seed * 7 + i, bias=True) for i, n in enumerate(widths)]
    for L_ in Ls:
        L_.setdefault("act_order", False)
    mods = [_module(L_, bits, gs, dtype, tiled) for L_ in Ls]
    for M in rnd.sample((1, 2, 4, 5, 8, 16, 40, 64, 128, 130, 300), 4):
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
        with torch.no_grad():
            ys = forward_multi(mods, x.to(DEV))
        for L_, y in zip(Ls, ys):
            y64 = O.forward_f64(x, L_["qweight"], L_["qzeros"], L_["scales"], L_["g_idx"], L_["bias"], bits, O.ZERO_WRAP)
            _close(y, y64, dtype, K, f"multi int{bits} g{gs} K={K} widths={widths} M={M} {dtype} copy={tiled}")


@pytest.mark.parametrize("seed", range(80))
def test_random_forced_plans_are_correct_or_refused(seed):
    """tuning structs drawn at random (kernel family, strip width, waves, K slices): a forced plan either runs and matches the oracle, or the call is REFUSED
    with a GptqError (GPTQ_ERR_UNSUPPORTED / _WORKSPACE: 'does not fit' is an error, never a silent fallback and never a wrong answer)."""
    rnd = random.Random(7000 + seed)
    bits, dtype, K, N, gs, act, tiled, epi, Ms = _draw(rnd, False)
    if dtype == torch.float32:
        dtype = torch.float16
    L = O.random_quant_layer(K, N, bits, gs, act_order=act, dtype=dtype, seed=seed, bias=True)
    L.setdefault("act_order", act)
    zm = "nowrap" if act else rnd.choice(("wrap", "nowrap"))
    try:
        q = _module(L, bits, gs, dtype, tiled, "none", zm)
    except (_lib.GptqError, ValueError) as e:
        pytest.skip(f"refused at post_init: {e}")
    W64 = _w64(L, bits, O.ZERO_NOWRAP if zm == "nowrap" else O.ZERO_WRAP)
    ran = 0
    for M in Ms[:4]:
        t = _lib.GptqTuning()
        t.path = rnd.choice((0, 1, 3, 5, 6, 8))
        t.lanes_n = rnd.choice((0, 0, 4, 8, 16))
        t.waves = rnd.choice((0, 0, 2, 4, 8, 16))
        t.ksplit = rnd.choice((0, 0, 1, 2, 4))
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(seed + M)) - 0.5).to(dtype)
        y64 = x.double() @ W64 + L["bias"].double()
        try:
            with torch.no_grad():
                y = q(x.to(DEV), tuning=t)
        except _lib.GptqError:
            continue
        ran += 1
        _close(y, y64, dtype, K, f"forced path={t.path} ln={t.lanes_n} waves={t.waves} ks={t.ksplit}: int{bits} g{gs} {K}x{N} M={M} {dtype} act={act} copy={tiled}")
    SEEN["forced_ran"] = SEEN.get("forced_ran", 0) + ran


def test_the_fuzz_reached_every_kernel_family():
    """Runs last in this file: the default plans above must have landed on the decode-copy kernel, the batched-decode and panel kernels, the fp32-math GEMV,
    a matrix-core GEMV on the checkpoint rows and at least one MFMA GEMM -- otherwise the draw has drifted away from what it is meant to cover."""
    if not (set(SEEN) - {"forced_ran"}):
        pytest.skip("run together with the fuzz cases")
    need = {"strips", "generic", "rows"}
    assert need <= set(SEEN), SEEN
    assert {"mfma", "mfma_generic", "stream"} & set(SEEN), SEEN
    assert {"tiled", "wide_sk", "panel", "mid", "stream64", "skinny64", "strip16", "f32_mfma"} & set(SEEN), SEEN
    assert SEEN.get("forced_ran", 1) > 0, SEEN
