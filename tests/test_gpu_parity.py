"""GPU (-m gpu): parity of the HIP path (through the C ABI) against the oracle, the reference's
known-answer vectors and the reference-generated fixtures.

Bars:  integer unpack / pack / dequant -> bit-exact.
       matmul -> |y - y_ref| <= atol + rtol*|y_ref| with the tolerances written at each test; the HIP
       kernels accumulate in fp32 (never fp16), so they are also required to be at least as close to
       the exact-math (fp64) result as the reference's own fp16 CPU path is.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import QuantLinear
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"

# fp tolerance of the matmul, per I/O dtype: (rtol, atol_per_sqrt(K/1024) relative to output scale)
TOL = {torch.float32: (1e-4, 1e-5), torch.float16: (2e-3, 2e-3), torch.bfloat16: (1.6e-2, 1.6e-2)}


@pytest.fixture(autouse=True)
def _checkpoint_layout_kernels():
    """This file pins the kernels that stream the CHECKPOINT layout (every packing, act-order, fp32, the forced-geometry grids and the planner's
    documented choices among them): layers are built without the round-4 decode copy, which has its own files (test_gpu_tiled.py and, through the
    default plan, test_gpu_baseline_configs.py)."""
    old = QuantLinear.TILED_DECODE
    QuantLinear.TILED_DECODE = False
    yield
    QuantLinear.TILED_DECODE = old


def _module_from(qweight, qzeros, scales, g_idx, bias, bits, group_size, zero_mode="auto"):
    K = qweight.shape[0] * 32 // bits
    N = qweight.shape[1]
    q = QuantLinear(bits, group_size, K, N, bias is not None, weight_dtype=scales.dtype, zero_mode=zero_mode)
    q.qweight, q.qzeros, q.scales = qweight.clone(), qzeros.clone(), scales.clone()
    if g_idx is not None:
        q.g_idx = g_idx.clone().to(torch.int32)
    if bias is not None:
        q.bias = bias.clone()
    return q.to(DEV)


def _zm(c):
    """zero_mode for a reference fixture: 'auto' resolves to the cuda_old convention for sequential
    g_idx; fixtures produced by the act-order class (qlinear_cuda.py) need its convention spelled out."""
    return "nowrap" if c.desc_act_class else "auto"


def _tuning(**kw):
    t = _lib.GptqTuning()
    for k, v in kw.items():
        setattr(t, k, v)
    return t


def _assert_close(y, ref, y64, dtype, K, what=""):
    rtol, atol = TOL[dtype]
    scale = max(1.0, float(y64.abs().max())) if y64 is not None else 1.0
    a = atol * scale * max(1.0, (K / 1024) ** 0.5)
    yf, rf = y.double().cpu(), ref.double().cpu()
    bad = (yf - rf).abs() > a + rtol * rf.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())} / {bad.numel()} out of tolerance, max abs diff {float((yf - rf).abs().max())}"


# ------------------------------------------------------------------------------------------- KATs
@pytest.mark.parametrize("fname,path", [
    ("kat_cuda_old_reference_1024.npz", 0), ("kat_cuda_old_reference_1024.npz", 1),
    ("kat_reference_old_half_256.npz", 0), ("kat_reference_old_half_256.npz", 1),
    ("kat_reference_old_no_half_256.npz", 0),
])
def test_known_answer_vectors(golden_dir, fname, path):
    """The reference's own golden vectors (tests/test_q4.py:1060-1122, 1752-1802, 1899-1941) with the
    reference's own tolerances."""
    import os
    z = np.load(os.path.join(golden_dir, fname))
    k, n = int(z["k"]), int(z["n"])
    dtype = {"float16": torch.float16, "float32": torch.float32}[str(z["dtype"])]
    qweight, qzeros, scales, x = O.golden_recipe_inputs(k, n, dtype=dtype)
    q = _module_from(qweight, qzeros, scales, None, None, 4, 128)
    with torch.no_grad():
        y = q(x.to(DEV), tuning=_tuning(path=path))[0][0].cpu()
    ref = torch.from_numpy(z["y"]).to(dtype)
    assert torch.allclose(y, ref, rtol=float(z["rtol"]), atol=max(float(z["atol"]), 1e-8)), (y - ref).abs().max()


# ------------------------------------------------------------------------------ integer / dequant
def test_unpack_bit_exact(ref_case):
    c = ref_case
    lib = _lib.load()
    qw, qz = c.qweight.to(DEV), c.qzeros.to(DEV)
    w = torch.empty((c.K, c.N), dtype=torch.uint8, device=DEV)
    _lib.check(lib.gptq_unpack_weights(qw.data_ptr(), c.K, c.N, c.bits, w.data_ptr(), None))
    G = c.qzeros.shape[0]
    for mode in (O.ZERO_WRAP, O.ZERO_NOWRAP):
        zt = torch.empty((G, c.N), dtype=torch.int32, device=DEV)
        _lib.check(lib.gptq_unpack_zeros(qz.data_ptr(), G, c.N, c.bits, int(mode == O.ZERO_NOWRAP), zt.data_ptr(), None))
        torch.cuda.synchronize()
        assert np.array_equal(zt.cpu().numpy(), O.unpack_zeros(c.qzeros, c.bits, mode))
    torch.cuda.synchronize()
    assert np.array_equal(w.cpu().numpy().astype(np.uint16), O.unpack_weights(c.qweight, c.bits))


@pytest.mark.parametrize("bits", [2, 3, 4, 8])
def test_unpack_bit_exact_random_words(bits):
    L = O.random_quant_layer(512, 384, bits, 64, seed=10 + bits)
    lib = _lib.load()
    w = torch.empty((512, 384), dtype=torch.uint8, device=DEV)
    _lib.check(lib.gptq_unpack_weights(L["qweight"].to(DEV).data_ptr(), 512, 384, bits, w.data_ptr(), None))
    torch.cuda.synchronize()
    assert np.array_equal(w.cpu().numpy().astype(np.uint16), O.unpack_weights(L["qweight"], bits))


def test_dequant_bit_exact_vs_reference(ref_case):
    """gptq_dequant == the reference's `weights` tensor, bit for bit (fixtures produced by pushing an
    identity through the reference forward)."""
    c = ref_case
    q = _module_from(c.qweight, c.qzeros, c.scales, c.g_idx, None, c.bits, c.group_size, zero_mode=_zm(c))
    W = q.dequantize().cpu()
    mode = O.reference_zero_mode(c.desc_act_class, c.bits)
    assert q.resolved_zero_mode() == int(mode == O.ZERO_NOWRAP)
    assert torch.equal(W, O.dequantize(c.qweight, c.qzeros, c.scales, c.g_idx, c.bits, mode))
    if c.bias is None:
        assert torch.equal(W, c.Wdq)


@pytest.mark.parametrize("bits,dtype", [(4, torch.float16), (3, torch.float16), (8, torch.bfloat16), (2, torch.float32)])
@pytest.mark.parametrize("act", [False, True])
def test_dequant_bit_exact_random_words(bits, dtype, act):
    L = O.random_quant_layer(512, 256, bits, 32, act_order=act, dtype=dtype, seed=3)
    for zm, mode in (("wrap", O.ZERO_WRAP), ("nowrap", O.ZERO_NOWRAP)):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, bits, 32, zero_mode=zm)
        assert torch.equal(q.dequantize().cpu(), O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], bits, mode))


# ------------------------------------------------------------------------------------ forward
@pytest.mark.parametrize("path", [0, 1])
def test_forward_matches_reference_fixture(ref_case, path):
    c = ref_case
    q = _module_from(c.qweight, c.qzeros, c.scales, c.g_idx, c.bias, c.bits, c.group_size, zero_mode=_zm(c))
    with torch.no_grad():
        y = q(c.x.to(DEV), tuning=_tuning(path=path))
    assert y.dtype == c.dtype and tuple(y.shape) == tuple(c.y.shape)
    mode = O.reference_zero_mode(c.desc_act_class, c.bits)
    y64 = O.forward_f64(c.x, c.qweight, c.qzeros, c.scales, c.g_idx, c.bias, c.bits, mode)
    _assert_close(y, c.y, y64, c.dtype, c.K, f"{c.name} vs reference y")
    # accumulating in fp32, the kernel must sit at least as close to exact math as the reference does
    err_k = (y.double().cpu() - y64).abs().max()
    err_r = (c.y.double() - y64).abs().max()
    assert err_k <= err_r * 3 + 2e-6 * max(1.0, float(y64.abs().max())), (float(err_k), float(err_r))


CASES = [
    # bits, gs, K, N, act, dtype, M
    (4, 128, 1024, 1024, False, torch.float16, 1),      # BASELINE config 1 (parity gate shape)
    (4, 128, 1024, 1024, False, torch.float16, 3),
    (4, 128, 2048, 512, False, torch.float16, 8),
    (4, 128, 1024, 1024, True, torch.float16, 1),       # act-order, re-sequenced fast path
    (4, 128, 1024, 1024, True, torch.float16, 5),
    (4, 32, 512, 768, False, torch.float16, 2),
    (4, 64, 512, 96, False, torch.bfloat16, 2),
    (4, 128, 512, 256, False, torch.float32, 2),
    (3, 32, 1024, 512, False, torch.float16, 1),        # BASELINE config 5 flavours
    (8, 32, 1024, 512, False, torch.float16, 1),
    (3, 32, 512, 256, True, torch.float16, 2),
    (8, 32, 512, 256, True, torch.float16, 2),
    (2, 64, 512, 256, False, torch.float16, 2),
    (2, 64, 512, 256, True, torch.bfloat16, 3),
    (4, 1024, 1024, 256, False, torch.float16, 1),      # group_size = -1 (one group)
    (4, 16, 256, 128, False, torch.float16, 2),         # group smaller than a 3-bit unit, fine for 4-bit
    (3, 16, 256, 128, False, torch.float16, 2),         # 3-bit unit (32 k) spans two groups -> per-k path
]


@pytest.mark.parametrize("bits,gs,K,N,act,dtype,M", CASES)
def test_forward_random_words_vs_oracle(bits, gs, K, N, act, dtype, M):
    L = O.random_quant_layer(K, N, bits, gs, act_order=act, dtype=dtype, seed=K + N + bits, bias=True)
    gen = torch.Generator().manual_seed(7)
    x = (torch.rand(M, K, generator=gen) - 0.5).to(dtype)
    for zm, mode in (("wrap", O.ZERO_WRAP), ("nowrap", O.ZERO_NOWRAP)):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], bits, gs, zero_mode=zm)
        yref = O.forward(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], bits, mode)
        y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], bits, mode)
        for path in (0, 1):
            with torch.no_grad():
                y = q(x.to(DEV), tuning=_tuning(path=path))
            _assert_close(y, yref, y64, dtype, K, f"path={path} zero={zm}")
            _assert_close(y, y64, y64, dtype, K, f"path={path} zero={zm} vs f64")


@pytest.mark.parametrize("bits,gs,K,N", [(3, 32, 1024, 512), (3, 128, 256, 64), (3, 128, 2048, 160), (3, 64, 11008, 96),
                                          (8, 32, 1024, 512), (8, 128, 512, 64), (8, 64, 2048, 160), (8, 32, 11008, 96)])
@pytest.mark.parametrize("M", [1, 2, 3, 4, 5, 8])
def test_magic_number_decode_3_and_8_bit(bits, gs, K, N, M):
    """3- / 8-bit fp16 matrix-core GEMV with the packed magic-number field decode (default) == the field-by-field form
    (tuning.reserved[1] = 1) == the oracle, for every row-tile instantiation (M = 1, 2, 3..4, two passes at 5..8 where the planner
    keeps the GEMV), both zero conventions, K tails (11008 = 344 units: the last pass of a workgroup is partly dead); one-hot rows
    of x return the dequantised rows bit for bit.  M = 2 is the case that caught a WAW hazard between an in-flight MFMA's write of a
    dead accumulator half and a VALU write hidden in inline asm (csrc/gemv.hip, MagicConsts)."""
    L = O.random_quant_layer(K, N, bits, gs, act_order=False, dtype=torch.float16, seed=3 * K + N + bits + M, bias=True)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half()
    hot = torch.zeros(M, K, dtype=torch.float16)
    rows = [(37 * (m + 1) + K // 2 * (m & 1)) % K for m in range(M)]
    for m, r in enumerate(rows):
        hot[m, r] = 1.0
    for zm, mode in (("wrap", O.ZERO_WRAP), ("nowrap", O.ZERO_NOWRAP)):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], bits, gs, zero_mode=zm)
        y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], bits, mode)
        t_old = _tuning(path=5)
        t_old.reserved[_lib.LAB.OPT] = _lib.LAB.OPT_FIELD_DECODE
        with torch.no_grad():
            y_new = q(x.to(DEV), tuning=_tuning(path=5))
            y_old = q(x.to(DEV), tuning=t_old)
            h_new = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, bits, gs, zero_mode=zm)(hot.to(DEV), tuning=_tuning(path=5))
        _assert_close(y_new, y64, y64, torch.float16, K, f"magic zero={zm} vs f64")
        _assert_close(y_old, y64, y64, torch.float16, K, f"field-by-field zero={zm} vs f64")
        _assert_close(y_new, y_old, y64, torch.float16, K, f"magic vs field-by-field zero={zm}")
        W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], bits, mode)             # [K, N] fp16, one rounding
        assert torch.equal(h_new.cpu(), torch.stack([W[r] for r in rows])), f"one-hot rows zero={zm}"


@pytest.mark.parametrize("ln,waves,ksplit", [(4, 16, 1), (4, 4, 1), (8, 8, 2), (16, 16, 4), (64, 4, 8), (64, 1, 1), (4, 2, 3)])
@pytest.mark.parametrize("path", [1, 5])
def test_forward_launch_shapes_agree(ln, waves, ksplit, path):
    """Every launch shape (strip width, waves, K split) of both checkpoint-layout GEMV kernels (1: fp32 math, 5: matrix core) gives the same answer
    (up to fp32 summation order) and is run-to-run bit-reproducible.  Round 6 cut the instantiations to what a plan can ask for: the fp32-math kernel has
    strips of 16 or 64 columns (lanes_n = 4 / 16), the matrix-core kernel up to 64 (lanes_n = 4 / 8 / 16); everything else -- and round 1's comparison
    kernels, tuning.path = 2 / 4 -- is refused with GPTQ_ERR_UNSUPPORTED, never silently replaced."""
    K, N, M = 2048, 1024, 2
    L = O.random_quant_layer(K, N, 4, 128, seed=5, bias=True)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(1)) - 0.5).half()
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, 128)
    y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP)
    t = _tuning(lanes_n=ln, waves=waves, ksplit=ksplit, path=path)
    if ln not in ((4, 16) if path == 1 else (4, 8, 16)):
        with pytest.raises(_lib.GptqError, match="lanes_n"):
            q(x.to(DEV), tuning=t)
        return
    with torch.no_grad():
        y1 = q(x.to(DEV), tuning=t)
        y2 = q(x.to(DEV), tuning=t)
    assert torch.equal(y1, y2)
    _assert_close(y1, y64, y64, torch.float16, K, f"ln={ln} waves={waves} ksplit={ksplit}")
    for retired in (2, 4):
        with pytest.raises(_lib.GptqError, match="retired"):
            q(x.to(DEV), tuning=_tuning(path=retired))


def test_act_order_resequencing_is_bit_exact():
    """qweight_seq (derived at post_init) holds exactly the rows of qweight in group-sorted order and
    leaves the checkpoint tensor untouched (the reference's exllama path overwrites it in place)."""
    K, N, bits, gs = 1024, 256, 4, 128
    L = O.random_quant_layer(K, N, bits, gs, act_order=True, seed=9)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, bits, gs)
    q.post_init()
    assert q.act_order and q._layer.qweight_seq
    qweight_seq, perm = q._keepalive[5], q._keepalive[6]
    assert np.array_equal(perm.cpu().numpy(), O.sequential_permutation(L["g_idx"]))
    w = O.unpack_weights(L["qweight"], bits)
    w_seq = O.unpack_weights(qweight_seq.cpu(), bits)
    assert np.array_equal(w_seq, w[perm.cpu().numpy()])
    assert torch.equal(q.qweight.cpu(), L["qweight"])
    for b in (2, 3, 8):
        Lb = O.random_quant_layer(512, 128, b, 32, act_order=True, seed=b)
        qb = _module_from(Lb["qweight"], Lb["qzeros"], Lb["scales"], Lb["g_idx"], None, b, 32)
        qb.post_init()
        p = qb._keepalive[6].cpu().numpy()
        assert np.array_equal(O.unpack_weights(qb._keepalive[5].cpu(), b), O.unpack_weights(Lb["qweight"], b)[p])


def test_device_pack_bit_exact(ref_case):
    """QuantLinear.pack on the GPU (gptq_pack_weights / gptq_pack_zeros) == reference pack()."""
    c = ref_case
    lin = torch.nn.Linear(c.K, c.N, bias=c.lin_bias is not None)
    lin.weight.data = c.W.to(c.dtype)
    if c.lin_bias is not None:
        lin.bias.data = c.lin_bias.clone()
    q = QuantLinear(c.bits, c.group_size, c.K, c.N, c.lin_bias is not None, weight_dtype=c.dtype, zero_mode=_zm(c))
    q.pack(lin, c.scale.to(c.qparams_dtype), c.zero.to(c.qparams_dtype), c.g_idx.clone())
    assert torch.equal(q.qweight.cpu(), c.qweight)
    assert torch.equal(q.qzeros.cpu(), c.qzeros)
    assert torch.equal(q.scales.cpu(), c.scales)
    # pack -> forward round trip reproduces the reference output
    q = q.to(DEV)
    with torch.no_grad():
        y = q(c.x.to(DEV))
    mode = O.reference_zero_mode(c.desc_act_class, c.bits)
    y64 = O.forward_f64(c.x, c.qweight, c.qzeros, c.scales, c.g_idx, c.bias, c.bits, mode)
    _assert_close(y, c.y, y64, c.dtype, c.K, "pack->forward")


def test_permute_columns_and_errors():
    lib = _lib.load()
    x = torch.randn(5, 256, device=DEV).half()
    perm = torch.randperm(256, device=DEV).to(torch.int32)
    out = torch.empty_like(x)
    _lib.check(lib.gptq_permute_columns(x.data_ptr(), perm.data_ptr(), 5, 256, 0, out.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(out, x[:, perm.long()])
    q = QuantLinear(4, 128, 256, 64, False).to(DEV)
    with pytest.raises(RuntimeError, match="features"):
        q(torch.zeros(1, 128, dtype=torch.float16, device=DEV))
    assert q(torch.zeros(0, 256, dtype=torch.float16, device=DEV)).shape == (0, 64)   # empty batch
    y = q(torch.zeros(2, 3, 256, dtype=torch.float32, device=DEV))                      # dtype cast + 3-D
    assert y.shape == (2, 3, 64) and y.dtype == torch.float32


# ---------------------------------------------------- BASELINE full sizes: size-independent properties
FULL = [(4096, 4096), (4096, 11008), (11008, 4096), (3584, 8192), (8192, 1024),   # + Llama-2-70B TP=8 shards (32-column strips / K split)
        (5120, 5120), (13824, 5120), (8192, 16384)]          # + Llama-13B projections and a 64-column-strip case: streamed kernel by default


@pytest.mark.parametrize("K,N", FULL)
def test_full_size_decode_properties(K, N):
    """Llama-7B shapes (BASELINE config 2) and larger, M=1: (a) EVERY output against x (fp64) @ W_oracle (fp64) with the tight full-size
    tolerance of tests/test_gpu_baseline_configs.py (rtol = atol = 1e-3 of the output scale, no sqrt(K) allowance),
    (b) linearity  f(a*x1 + x2) = a*f(x1) + f(x2)  within fp16 rounding, (c) column-slice consistency: the layer restricted to
    columns [n0,n1) gives the same outputs (the out_features sharding used for TP), (d) bit reproducibility."""
    L = O.random_quant_layer(K, N, 4, 128, seed=K // 7 + N)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], None, None, 4, 128)
    gen = torch.Generator().manual_seed(3)
    x1 = (torch.rand(1, K, generator=gen) - 0.5).half()
    x2 = (torch.rand(1, K, generator=gen) - 0.5).half()
    with torch.no_grad():
        y1, y1b = q(x1.to(DEV)), q(x1.to(DEV))
        y2 = q(x2.to(DEV))
        y3 = q((0.5 * x1 + x2).to(DEV))
    assert torch.equal(y1, y1b)
    # (a) every output against the oracle's dequantised weight, multiplied in fp64
    W64 = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 4, O.ZERO_WRAP).to(DEV).double()
    ref = x1.to(DEV).double() @ W64
    scale64 = float(ref.abs().max())
    bad = (y1.double() - ref).abs() > 1e-3 * scale64 + 1e-3 * ref.abs()
    assert not bool(bad.any()), f"{K}x{N} M=1: {int(bad.sum())}/{bad.numel()} outputs out of tolerance, first at {torch.nonzero(bad)[0].tolist()}"
    n0 = (N // 2) // 32 * 32
    sl = slice(n0, n0 + 256)
    y64 = ref[:, sl].cpu()
    del W64
    # (b) linearity (x combination is rounded to fp16 -> compare loosely, relative to output scale)
    lin = 0.5 * y1.float() + y2.float()
    scale = float(lin.abs().max())
    assert float((y3.float() - lin).abs().max()) <= 6e-3 * scale
    # (c) column slice as its own layer
    qs = _module_from(L["qweight"][:, sl].contiguous(), L["qzeros"][:, n0 // 8:(n0 + 256) // 8].contiguous(),
                      L["scales"][:, sl].contiguous(), None, None, 4, 128)
    with torch.no_grad():
        ys = qs(x1.to(DEV))
    badc = (ys.double().cpu() - y64).abs() > 1e-3 * scale64 + 1e-3 * y64.abs()          # the sliced layer against the same fp64 reference, same tolerance
    assert not bool(badc.any()), f"column-sliced layer {K}x{N}: {int(badc.sum())}/{badc.numel()} outputs out of tolerance"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K,N,M,act", [(4096, 11008, 16, False), (11008, 4096, 16, True), (4096, 11008, 64, True), (11008, 4096, 40, False),
                                       (8192, 3584, 8, False), (4096, 4096, 16, False)])
def test_full_size_batched_decode_properties(K, N, M, act, dtype):
    """Batched decode (4 < M <= 64) at the BASELINE shapes with the DEFAULT plan -- the streamed 64-column-strip kernel without
    K split (172 strips), with 4 in-launch K slices (64 / 56 strips), on an act-order layer (permuted x + qweight_seq); 4096^2
    keeps the 16-column strips: (a) fp64 oracle on a middle and a ragged right-edge column slice, (b) one-hot rows return the
    exact dequantised rows, (c) bit reproducibility, (d) the plan is the documented one."""
    L = O.random_quant_layer(K, N, 4, 128, seed=K // 7 + N + M, act_order=act, dtype=dtype, bias=True)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"] if act else None, L["bias"], 4, 128)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
    with torch.no_grad():
        y, yb = q(x.to(DEV)), q(x.to(DEV))
    assert torch.equal(y, yb)
    d = _lib.describe_plan(q._layer, M)
    assert d["kernel"] == ("strip16" if (K, N) == (4096, 4096) else ("mid" if M > 16 else "stream64")), d
    if d["kernel"] == "stream64":
        assert d["ksplit"] == (1 if N >= 10240 else 4), d
    if d["kernel"] == "mid":                         # 4096-wide layers: 2 row blocks x 2 K slices at 33..64 rows; 172 strips: neither
        assert (d["tiles"], d["ksplit"]) == (("1x172", 1) if N >= 10240 else ("2x64", 2)), d
    mode = O.reference_zero_mode(act, 4)
    for n0 in ((N // 2) // 32 * 32, N - 96):
        n1 = min(N, n0 + 96)
        sl = slice(n0, n1)
        y64 = O.forward_f64(x, L["qweight"][:, sl], L["qzeros"][:, n0 // 8:n1 // 8], L["scales"][:, sl], L["g_idx"] if act else None,
                            L["bias"][sl], 4, mode)
        _assert_close(y[:, sl], y64, y64, dtype, K, f"batched decode full size, columns {n0}:{n1}")
    ks = (torch.arange(M) * 977 + 13) % K
    xo = torch.zeros(M, K, dtype=dtype)
    xo[torch.arange(M), ks] = 1.0
    with torch.no_grad():
        yo = q(xo.to(DEV)).cpu()
    sl = slice(N - 64, N)
    W = O.dequantize(L["qweight"][:, sl], L["qzeros"][:, (N - 64) // 8:], L["scales"][:, sl], L["g_idx"] if act else None, 4, mode)
    expect = (W[ks].float() + L["bias"][sl].float()).to(dtype)
    assert torch.equal(yo[:, sl], expect)


# ------------------------------------------------------------------------- 17 .. 128 rows: everything by LDS DMA (gemm_mid_kernel)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("stages,ksplit,xreg", [(0, 0, 0), (2, 1, 0), (3, 3, 0), (2, 8, 0), (3, 2, 1), (2, 5, 3), (2, 4, 10), (3, 1, 10)])
@pytest.mark.parametrize("M,K,N,gs,act", [(17, 1024, 128, 64, False), (32, 2048, 512, 128, True), (33, 4096, 1024, 128, False), (50, 4096, 1088, 128, True),
                                          (64, 224, 64, 32, False), (65, 1024, 256, 32, True), (96, 11008, 256, 128, False), (100, 2048, 192, 256, False),
                                          (128, 4096, 512, 128, True), (128, 512, 64, 512, False), (5, 512, 128, 128, False)])
def test_mid_kernel(M, K, N, gs, act, dtype, stages, ksplit, xreg):
    """17 .. 128 rows (and fewer, forced), 4-bit: gemm_mid_kernel (tuning.path = 3, reserved[2] = 5; the default in most of that range) against
    the fp64 oracle for default and forced launch geometries (stages in flight x in-launch K slices x the register variant of the x path; 2 / 4 / 6 /
    8 row tiles, ragged last row tile, K ranges that do not divide by waves, one K-step per group and one group per layer, act-order through the
    permuted-x pre-pass), with one-hot rows (the exact dequantised rows come back: catches any row / column / k-slot mix-up of the DMA lane
    layouts), twice (flags cleared) and bit-reproducible."""
    L = O.random_quant_layer(K, N, 4, gs, act_order=act, seed=M + K + N, bias=True, dtype=dtype)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, gs, zero_mode="wrap")
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
    y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP)
    t = _tuning(path=3, ksplit=ksplit)
    # xreg: 1 = x through registers, 3 = granule combine instead of flags, 10 = 128-column strips (two 64-column halves per wave; <= 64 rows, N % 128 == 0)
    if xreg == 10 and (M > 64 or N % 128):
        pytest.skip("128-column strips: up to 64 rows, N a multiple of 128")
    t.reserved[_lib.LAB.DEPTH], t.reserved[_lib.LAB.OPT], t.reserved[_lib.LAB.GEMM_KERNEL], t.reserved[_lib.LAB.GEMM_VARIANT] = stages, (xreg if xreg < 10 else 0), _lib.LAB.GEMM_MID, (2 if xreg == 10 else 0)
    q.post_init()
    d = _lib.describe_plan(q._layer, M, t)
    assert d["kernel"] == "mid", d
    if xreg == 10:
        assert d["tiles"] == f"1x{N // 128}", d
    with torch.no_grad():
        y, yb = q(x.to(DEV), tuning=t), q(x.to(DEV), tuning=t)
    assert torch.equal(y, yb)
    _assert_close(y, y64, y64, dtype, K, f"mid {d} vs f64")
    ks = (torch.arange(M) * 37 + 5) % K
    xo = torch.zeros(M, K, dtype=dtype)
    xo[torch.arange(M), ks] = 1.0
    with torch.no_grad():
        yo = q(xo.to(DEV), tuning=t).cpu()
    W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 4, O.ZERO_WRAP)
    expect = (W[ks].float() + L["bias"].float()).to(dtype)
    assert torch.equal(yo, expect)
    # the ticket half of the workspace header is zero again (flags cleared by the owner slices)
    from autogptq_amd.qlinear_mi355x import _WORKSPACE
    torch.cuda.synchronize()
    for ent in _WORKSPACE.values():
        if ent[0].numel() >= 65536:
            assert int(ent[0][:32768].count_nonzero()) == 0, "the ticket half of the header must be left zero by every launch"
            # granule combine: every consumed {fp32, tag} granule was cleared -- no word of the body carries the tag pattern (0x7FE.....) any more
            body = ent[0][65536:65536 + (ent[0].numel() - 65536) // 4 * 4].view(torch.int32)
            assert int(((body & -0x200000) == 0x7FE00000).sum()) == 0, "valid-looking granule tags left behind in the exchange area"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rbs,ksplit", [(0, 0), (1, 1), (2, 2), (1, 4), (4, 1)])
@pytest.mark.parametrize("M,K,N,gs,act", [(8, 1024, 128, 32, False), (17, 2048, 256, 32, True), (64, 4096, 512, 128, False), (100, 1024, 192, 64, True),
                                          (128, 11008, 128, 32, False), (200, 512, 64, 32, False)])
def test_mid_kernel_int8(M, K, N, gs, act, dtype, rbs, ksplit):
    """gemm_mid_kernel on 8-bit layers (tuning.path = 3, reserved[2] = 5): a K-step is 8 packed rows = two weight DMAs, the lane's two words per
    column come from rows 2 kg / 2 kg + 1, zero-points are bytes (both conventions), one group per K-step at group_size 32; row blocks and K slices
    as for 4-bit; fp64 oracle, one-hot rows, bit-reproducible, both zero conventions."""
    if rbs > (M + 15) // 16 or (M > 128 and rbs * 8 * 16 < M):
        pytest.skip("row blocks: at most one per row tile; 129+ rows need blocks of <= 8 row tiles")
    L = O.random_quant_layer(K, N, 8, gs, act_order=act, seed=M + K + N, bias=True, dtype=dtype)
    for zm in ("wrap", "nowrap"):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 8, gs, zero_mode=zm)
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
        mode = O.ZERO_WRAP if zm == "wrap" else O.ZERO_NOWRAP
        y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 8, mode)
        t = _tuning(path=3, ksplit=ksplit, lanes_n=rbs)
        t.reserved[_lib.LAB.GEMM_KERNEL] = _lib.LAB.GEMM_MID
        q.post_init()
        d = _lib.describe_plan(q._layer, M, t)
        if d["kernel"] != "mid":                     # one group per K-step and a long unsplit K: the per-wave group table does not fit the LDS
            assert gs == 32 and K >= 8192 and ksplit <= 1, d
            pytest.skip("group table too large for this geometry")
        with torch.no_grad():
            y, yb = q(x.to(DEV), tuning=t), q(x.to(DEV), tuning=t)
        assert torch.equal(y, yb)
        _assert_close(y, y64, y64, dtype, K, f"mid int8 {d} vs f64")
        ks = (torch.arange(M) * 37 + 5) % K
        xo = torch.zeros(M, K, dtype=dtype)
        xo[torch.arange(M), ks] = 1.0
        with torch.no_grad():
            yo = q(xo.to(DEV), tuning=t).cpu()
        W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 8, mode)
        assert torch.equal(yo, (W[ks].float() + L["bias"].float()).to(dtype))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rbs,ksplit", [(0, 0), (1, 1), (2, 2), (1, 4), (4, 1)])
@pytest.mark.parametrize("M,K,N,gs,act", [(8, 1024, 128, 32, False), (17, 2048, 256, 32, True), (64, 4096, 384, 128, False), (100, 1024, 640, 64, True),
                                          (128, 11008, 128, 32, False), (200, 512, 128, 32, False)])
def test_mid_kernel_int3(M, K, N, gs, act, dtype, rbs, ksplit):
    """gemm_mid_kernel on 3-bit layers: a K-step is three packed rows (one 48-lane weight DMA), every lane funnels its 24-bit window out of the 96-bit
    column stream, zero-points are a 24-byte run per strip that is 16-byte aligned only on even strips (N = 384 / 640: odd strips too); fp64 oracle,
    one-hot rows, bit-reproducible, both zero conventions."""
    if rbs > (M + 15) // 16 or (M > 128 and rbs * 8 * 16 < M):
        pytest.skip("row blocks: at most one per row tile; 129+ rows need blocks of <= 8 row tiles")
    L = O.random_quant_layer(K, N, 3, gs, act_order=act, seed=M + K + N, bias=True, dtype=dtype)
    for zm in ("wrap", "nowrap"):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 3, gs, zero_mode=zm)
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
        mode = O.ZERO_WRAP if zm == "wrap" else O.ZERO_NOWRAP
        y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 3, mode)
        t = _tuning(path=3, ksplit=ksplit, lanes_n=rbs)
        t.reserved[_lib.LAB.GEMM_KERNEL] = _lib.LAB.GEMM_MID
        q.post_init()
        d = _lib.describe_plan(q._layer, M, t)
        if d["kernel"] != "mid":
            assert gs == 32 and K >= 8192 and ksplit <= 1, d
            pytest.skip("group table too large for this geometry")
        with torch.no_grad():
            y, yb = q(x.to(DEV), tuning=t), q(x.to(DEV), tuning=t)
        assert torch.equal(y, yb)
        _assert_close(y, y64, y64, dtype, K, f"mid int3 {d} vs f64")
        ks = (torch.arange(M) * 37 + 5) % K
        xo = torch.zeros(M, K, dtype=dtype)
        xo[torch.arange(M), ks] = 1.0
        with torch.no_grad():
            yo = q(xo.to(DEV), tuning=t).cpu()
        W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 3, mode)
        assert torch.equal(yo, (W[ks].float() + L["bias"].float()).to(dtype))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rbs,ksplit", [(0, 0), (1, 1), (2, 2), (1, 4)])
@pytest.mark.parametrize("M,K,N,gs,act", [(8, 1024, 128, 32, False), (17, 2048, 256, 64, True), (64, 4096, 320, 128, False), (100, 1024, 192, 64, True), (128, 2048, 64, 2048, False)])
def test_mid_kernel_int2(M, K, N, gs, act, dtype, rbs, ksplit):
    """gemm_mid_kernel on 2-bit layers: a K-step is two packed rows (one 32-lane weight DMA), the lane's 8 values are 16 bits of a word spread over the
    halves of a register by one v_perm, four (v_and_or, packed fma) pairs; fp64 oracle, one-hot rows, bit-reproducible, both zero conventions."""
    if rbs > (M + 15) // 16:
        pytest.skip("row blocks: at most one per row tile")
    L = O.random_quant_layer(K, N, 2, gs, act_order=act, seed=M + K + N, bias=True, dtype=dtype)
    for zm in ("wrap", "nowrap"):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 2, gs, zero_mode=zm)
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
        mode = O.ZERO_WRAP if zm == "wrap" else O.ZERO_NOWRAP
        y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 2, mode)
        t = _tuning(path=3, ksplit=ksplit, lanes_n=rbs)
        t.reserved[_lib.LAB.GEMM_KERNEL] = _lib.LAB.GEMM_MID
        q.post_init()
        d = _lib.describe_plan(q._layer, M, t)
        assert d["kernel"] == "mid", d
        with torch.no_grad():
            y, yb = q(x.to(DEV), tuning=t), q(x.to(DEV), tuning=t)
        assert torch.equal(y, yb)
        _assert_close(y, y64, y64, dtype, K, f"mid int2 {d} vs f64")
        ks = (torch.arange(M) * 37 + 5) % K
        xo = torch.zeros(M, K, dtype=dtype)
        xo[torch.arange(M), ks] = 1.0
        with torch.no_grad():
            yo = q(xo.to(DEV), tuning=t).cpu()
        W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 2, mode)
        assert torch.equal(yo, (W[ks].float() + L["bias"].float()).to(dtype))


@pytest.mark.parametrize("bits,gs", [(8, 32), (3, 32), (8, 128), (3, 64)])
@pytest.mark.parametrize("M", [6, 16, 64])
def test_mid_multi_layer_launch_3_and_8_bit(bits, gs, M):
    """gptq_forward_multi on 3- / 8-bit layers that share x at 5..128 rows: one gemm_mid_kernel launch (forced and by default), every layer against the
    fp64 oracle and against its own single-layer forward."""
    from autogptq_amd.qlinear_mi355x import forward_multi
    K, widths = 2048, (512, 128, 1024)
    Ls = [O.random_quant_layer(K, n, bits, gs, seed=57 + i + M + bits, bias=(i == 1), dtype=torch.float16) for i, n in enumerate(widths)]
    qs = [_module_from(L["qweight"], L["qzeros"], L["scales"], None, L["bias"], bits, gs) for L in Ls]
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half().to(DEV)
    t = _tuning(path=3)
    t.reserved[_lib.LAB.GEMM_KERNEL] = _lib.LAB.GEMM_MID
    with torch.no_grad():
        yf = forward_multi(qs, x, tuning=t)
        yd = forward_multi(qs, x)
        sep = [q(x) for q in qs]
    mode = O.reference_zero_mode(False, bits)
    for y, d, s_, L in zip(yf, yd, sep, Ls):
        ref = O.forward_f64(x.cpu(), L["qweight"], L["qzeros"], L["scales"], None, L["bias"], bits, mode)
        _assert_close(y, ref, ref, torch.float16, K, "mid multi (forced) vs oracle")
        _assert_close(d, ref, ref, torch.float16, K, "mid multi (default) vs oracle")
        _assert_close(s_, ref, ref, torch.float16, K, "single layer vs oracle")


@pytest.mark.parametrize("rbs,ksplit", [(2, 1), (4, 2), (8, 1), (3, 4)])
@pytest.mark.parametrize("M,K,N,gs,act,dtype", [(33, 2048, 256, 128, False, torch.float16), (64, 4096, 512, 128, True, torch.float16), (100, 1024, 192, 32, False, torch.bfloat16),
                                               (128, 11008, 128, 128, False, torch.float16), (128, 4096, 1024, 64, True, torch.bfloat16),
                                               (200, 2048, 128, 128, False, torch.float16), (256, 1024, 256, 128, True, torch.bfloat16)])
def test_mid_kernel_row_blocks(M, K, N, gs, act, dtype, rbs, ksplit):
    """gemm_mid_kernel with workgroups along M (tuning.lanes_n = row blocks): each workgroup owns 16 rt rows of x, the (row block, strip) tiles
    have their own flag words, ragged last block; fp64 oracle, one-hot rows, bit-reproducible, header left zero."""
    L = O.random_quant_layer(K, N, 4, gs, act_order=act, seed=M + K + N + rbs, bias=True, dtype=dtype)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, gs, zero_mode="wrap")
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
    y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP)
    t = _tuning(path=3, ksplit=ksplit, lanes_n=rbs)
    t.reserved[_lib.LAB.GEMM_KERNEL] = _lib.LAB.GEMM_MID
    q.post_init()
    d = _lib.describe_plan(q._layer, M, t)
    assert d["kernel"] == "mid" and int(d["tiles"].split("x")[0]) > 1, d
    with torch.no_grad():
        y, yb = q(x.to(DEV), tuning=t), q(x.to(DEV), tuning=t)
    assert torch.equal(y, yb)
    _assert_close(y, y64, y64, dtype, K, f"mid row blocks {d} vs f64")
    ks = (torch.arange(M) * 37 + 5) % K
    xo = torch.zeros(M, K, dtype=dtype)
    xo[torch.arange(M), ks] = 1.0
    with torch.no_grad():
        yo = q(xo.to(DEV), tuning=t).cpu()
    W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 4, O.ZERO_WRAP)
    assert torch.equal(yo, (W[ks].float() + L["bias"].float()).to(dtype))
    from autogptq_amd.qlinear_mi355x import _WORKSPACE
    torch.cuda.synchronize()
    for ent in _WORKSPACE.values():
        if ent[0].numel() >= 65536:
            assert int(ent[0][:32768].count_nonzero()) == 0


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("stages,ksplit", [(0, 0), (2, 3), (3, 8)])
@pytest.mark.parametrize("M,K,widths,gs", [(17, 1024, (512, 192, 1024), 128), (40, 2048, (1088, 128), 64), (64, 4096, (256, 256, 256, 64), 128),
                                           (128, 512, (704, 704), 32), (96, 4096, (4096, 1024, 1024), 128)])
def test_mid_multi_layer_launch(M, K, widths, gs, dtype, stages, ksplit):
    """gptq_forward_multi at 17 .. 128 rows: 2..4 plain layers that share x in ONE gemm_mid_kernel launch (forced with tuning.path = 3 /
    reserved[2] = 5, and by default where the planner prefers it) -- every layer against the fp64 oracle and against its own single-layer forward,
    bias on some layers, flags cleared."""
    from autogptq_amd.qlinear_mi355x import forward_multi
    Ls = [O.random_quant_layer(K, n, 4, gs, seed=177 + i + M, bias=(i % 2 == 1), dtype=dtype) for i, n in enumerate(widths)]
    qs = [_module_from(L["qweight"], L["qzeros"], L["scales"], None, L["bias"], 4, gs) for L in Ls]
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype).to(DEV)
    t = _tuning(path=3, ksplit=ksplit)
    t.reserved[_lib.LAB.DEPTH], t.reserved[_lib.LAB.GEMM_KERNEL] = stages, _lib.LAB.GEMM_MID
    with torch.no_grad():
        ys = forward_multi(qs, x, tuning=t)
        ys2 = forward_multi(qs, x, tuning=t)
        yd = forward_multi(qs, x)
        sep = [q(x) for q in qs]
    mode = O.reference_zero_mode(False, 4)
    for y, y2, d, s_, L in zip(ys, ys2, yd, sep, Ls):
        assert torch.equal(y, y2)
        ref = O.forward_f64(x.cpu(), L["qweight"], L["qzeros"], L["scales"], None, L["bias"], 4, mode)
        _assert_close(y, ref, ref, dtype, K, "mid multi vs oracle")
        _assert_close(d, ref, ref, dtype, K, "forward_multi default vs oracle")
        _assert_close(y, s_, ref, dtype, K, "mid multi vs separate")


# ------------------------------------------------------------------- MFMA prefill path (gptq_gemm)
GEMM_CASES = [
    # bits, gs, K, N, act, dtype, M
    (4, 128, 1024, 1024, False, torch.float16, 128),    # BK=64 kernel, one full tile
    (4, 128, 1024, 512, False, torch.float16, 200),     # ragged M (2 row tiles, second partial)
    (4, 128, 512, 96, False, torch.float16, 9),         # N not a multiple of the 256-column tile, MT=1
    (4, 128, 512, 320, False, torch.float16, 33),       # MT=2, ragged N
    (4, 32, 512, 256, False, torch.float16, 64),        # group_size 32 -> BK=32 kernel
    (4, 128, 1024, 1024, True, torch.float16, 128),     # act-order: qweight_seq + permuted x
    (4, 128, 2048, 256, True, torch.float16, 70),
    (4, 64, 1024, 256, False, torch.bfloat16, 128),     # bf16: fp32-math dequant
    (4, 128, 1024, 256, True, torch.bfloat16, 40),
    (3, 32, 1024, 256, False, torch.float16, 128),      # 3-bit units straddle words
    (3, 64, 1024, 160, True, torch.float16, 48),
    (8, 32, 512, 256, False, torch.float16, 128),
    (8, 128, 1024, 256, True, torch.bfloat16, 20),
    (2, 64, 1024, 256, False, torch.float16, 128),
    (2, 32, 512, 128, True, torch.float16, 16),
    (2, 64, 2048, 512, False, torch.float16, 300),      # 2-bit fp16 on the tiled kernel (packed magic-number decode, round 3)
    (2, 128, 1024, 320, True, torch.float16, 520),
    (2, 64, 1024, 256, False, torch.bfloat16, 260),     # 2-bit bf16: field by field
    (4, 1024, 1024, 256, False, torch.float16, 96),     # one group for the whole layer
    (4, 32, 32 * 7, 64, False, torch.float16, 12),      # K = 224: odd number of K-steps
    (4, 128, 512, 8448, False, torch.float16, 6),       # M = 5..8 on a wide layer (N > 8192): auto dispatch takes the tiled kernel
    (4, 128, 1024, 8448, True, torch.float16, 8),
]


@pytest.mark.parametrize("bits,gs,K,N,act,dtype,M", GEMM_CASES)
def test_gemm_random_words_vs_oracle(bits, gs, K, N, act, dtype, M):
    """MFMA path forced (tuning.path = 3), both zero conventions, against the oracle's reference-order
    result and the exact (fp64) result.  The dequantised B fragments equal the reference's `weights`
    bit for bit, so the only difference left is fp32 MFMA accumulation vs ATen's CPU summation."""
    L = O.random_quant_layer(K, N, bits, gs, act_order=act, dtype=dtype, seed=K + N + bits + M, bias=True)
    gen = torch.Generator().manual_seed(11)
    x = (torch.rand(M, K, generator=gen) - 0.5).to(dtype)
    for zm, mode in (("wrap", O.ZERO_WRAP), ("nowrap", O.ZERO_NOWRAP)):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], bits, gs, zero_mode=zm)
        yref = O.forward(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], bits, mode)
        y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], bits, mode)
        with torch.no_grad():
            y = q(x.to(DEV), tuning=_tuning(path=3))
            y_auto = q(x.to(DEV))
        assert torch.equal(y, y_auto), "auto dispatch for M > 8 must take the MFMA path"
        _assert_close(y, yref, y64, dtype, K, f"gemm zero={zm}")
        _assert_close(y, y64, y64, dtype, K, f"gemm zero={zm} vs f64")


def test_gemm_transpose_detecting():
    """x = one-hot rows and an asymmetric weight: out[m, :] must be exactly the dequantised row k(m)
    (catches any row/column or k-slot mix-up in the MFMA fragment layouts)."""
    K, N, bits, gs = 256, 256, 4, 128
    L = O.random_quant_layer(K, N, bits, gs, seed=77)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, bits, gs)
    M = 160
    ks = torch.arange(M) * 37 % K
    x = torch.zeros(M, K, dtype=torch.float16)
    x[torch.arange(M), ks] = 1.0
    with torch.no_grad():
        y = q(x.to(DEV), tuning=_tuning(path=3)).cpu()
    W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], bits, O.ZERO_WRAP)
    assert torch.equal(y, W[ks])


@pytest.mark.parametrize("ksplit", [2, 3, 8])
def test_gemm_split_k_agrees_and_is_deterministic(ksplit):
    K, N, M = 2048, 512, 24
    L = O.random_quant_layer(K, N, 4, 128, seed=5, bias=True)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(1)) - 0.5).half()
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, 128)
    y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP)
    t = _tuning(ksplit=ksplit, path=3)
    with torch.no_grad():
        y1 = q(x.to(DEV), tuning=t)
        y2 = q(x.to(DEV), tuning=t)
    assert torch.equal(y1, y2)
    _assert_close(y1, y64, y64, torch.float16, K, f"ksplit={ksplit}")


@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("variant", [1, 2, 3])
def test_gemm_schedule_variants(variant, act):
    """tuning.reserved[3] selects an instruction schedule of the big-tile kernel (1 plain, 2 setprio, 3 cross-step
    pipeline).  Without act-order all of them accumulate the same products in the same order: bit-identical."""
    K, N, M = 1024, 512, 300
    L = O.random_quant_layer(K, N, 4, 128, act_order=act, seed=31 + variant, bias=True)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(3)) - 0.5).half().to(DEV)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, 128, zero_mode="wrap")
    t = _tuning(path=3)
    t.reserved[_lib.LAB.GEMM_VARIANT] = variant
    t6 = _tuning(path=3)
    t6.reserved[_lib.LAB.GEMM_VARIANT] = _lib.LAB.VARIANT_ONE_K_GROUP
    with torch.no_grad():
        y0 = q(x, tuning=t6)
        y1 = q(x, tuning=t)
    if not act:
        assert torch.equal(y0, y1)
    y64 = O.forward_f64(x.cpu(), L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP)
    _assert_close(y1, y64, y64, torch.float16, K, f"variant {variant}")


@pytest.mark.parametrize("dtype,act,K,N,M,ksplit", [
    (torch.float16, False, 1024, 512, 300, 0), (torch.float16, True, 1024, 512, 300, 0), (torch.bfloat16, False, 2048, 256, 130, 0),
    (torch.bfloat16, True, 1024, 256, 128, 0), (torch.float16, False, 2048, 320, 97, 2), (torch.float16, True, 4096, 4096, 2048, 0),
    (torch.float16, False, 4096, 4096, 1500, 0)])
def test_gemm_k_groups_inside_workgroup(dtype, act, K, N, M, ksplit):
    """8-wave form of the big-tile kernel (two K halves per workgroup summed through LDS; the planner picks it when a launch
    has at most one tile per CU, reserved[3] = 7 / 6 force it on / off): against the fp64 oracle on a row/column sample and
    against the 4-wave form everywhere; bit-reproducible."""
    L = O.random_quant_layer(K, N, 4, 128, act_order=act, seed=K + N + M, bias=True, dtype=dtype)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(4)) - 0.5).to(dtype).to(DEV)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, 128, zero_mode="wrap")
    t6, t7 = _tuning(path=3, ksplit=ksplit), _tuning(path=3, ksplit=ksplit)
    t6.reserved[_lib.LAB.GEMM_VARIANT], t7.reserved[_lib.LAB.GEMM_VARIANT] = _lib.LAB.VARIANT_ONE_K_GROUP, _lib.LAB.VARIANT_TWO_K_GROUPS
    with torch.no_grad():
        y6 = q(x, tuning=t6)
        y7 = q(x, tuning=t7)
        y7b = q(x, tuning=t7)
        y_auto = q(x)
    assert torch.equal(y7, y7b)
    if ksplit == 0:
        assert torch.equal(y_auto, y7), "at most one tile per CU here: auto dispatch must pick the 8-wave form"
    rows = torch.arange(0, M, max(1, M // 61))
    cols = slice(0, min(N, 512))
    y64 = O.forward_f64(x[rows].cpu(), L["qweight"][:, cols], L["qzeros"][:, : cols.stop * 4 // 32], L["scales"][:, cols], L["g_idx"],
                        L["bias"][cols], 4, O.ZERO_WRAP)
    _assert_close(y7[rows][:, cols], y64, y64, dtype, K, "8-wave vs f64")
    _assert_close(y7, y6.double(), y64, dtype, K, "8-wave vs 4-wave")


def _check_balanced_tail(bits, gs, dtype, act, K, N, M, slices=None):
    from autogptq_amd import qlinear_mi355x as QM
    L = O.random_quant_layer(K, N, bits, gs, act_order=act, seed=K + N + M + bits, bias=True, dtype=dtype)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(40)) - 0.5).to(dtype).to(DEV)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], bits, gs, zero_mode="wrap")
    t_on, t_off = _tuning(path=3), _tuning(path=3)
    t_on.reserved[_lib.LAB.GEMM_VARIANT], t_off.reserved[_lib.LAB.GEMM_VARIANT] = _lib.LAB.VARIANT_TAIL_ON, _lib.LAB.VARIANT_TAIL_OFF
    with torch.no_grad():
        y_off = q(x, tuning=t_off)
        ys = [q(x, tuning=t_on) for _ in range(4)]
        y_auto = q(x)
    plan = _lib.describe_plan(q._layer, M, t_on)
    tiles = ((M + 127) // 128) * (N // 256)
    assert plan["kernel"] == "tiled" and plan["tail"] == tiles % 256 and plan["tail_slices"] in ((slices,) if slices else (2, 4, 8)), plan
    assert _lib.describe_plan(q._layer, M, t_off)["tail"] == 0
    for y in ys[1:]:
        assert torch.equal(ys[0], y)
    assert torch.equal(ys[0], y_auto), "the rule is the default"
    torch.cuda.synchronize()
    for ent in QM._WORKSPACE.values():
        hdr = ent[0][:_lib.WS_HEADER_BYTES].view(torch.int32)
        assert int(hdr[:8192].abs().max().item()) == 0, "ticket / flag words must be zero between launches"
        assert int(hdr[_lib.WS_HEADER_BYTES // 4 - 16 + 2].item()) == 0, "a bounded wait gave up"
    rows = torch.cat([torch.arange(0, M, max(1, M // 53)), torch.arange(max(0, M - 130), M, 7)]).unique()
    cols = slice(N - 512, N)                      # the last column tiles: the split tiles sit at the end of the tile order
    y64 = O.forward_f64(x[rows].cpu(), L["qweight"][:, cols], L["qzeros"][:, cols.start * bits // 32: cols.stop * bits // 32], L["scales"][:, cols],
                        L["g_idx"], L["bias"][cols], bits, O.ZERO_WRAP)
    _assert_close(ys[0][rows][:, cols], y64, y64, dtype, K, "balanced tail vs f64")
    _assert_close(ys[0], y_off.double(), y64, dtype, K, "balanced tail vs whole tiles")


@pytest.mark.parametrize("dtype,act,K,N,M,slices", [
    (torch.float16, False, 512, 2304, 7424, 2), (torch.float16, True, 512, 2304, 7300, 2), (torch.bfloat16, False, 2048, 2304, 7424, 4),
    (torch.bfloat16, True, 1024, 2304, 3800, 2), (torch.float16, True, 4096, 11008, 768, 8), (torch.float16, False, 4096, 4096, 2176, 4)])
def test_tiled_gemm_balanced_tail(dtype, act, K, N, M, slices):
    """Balanced tail of the big-tile kernel (the default; tuning.reserved[3] = 40 / 41 = the planner's rule / off): the tiles past the last full round
    of 256 run as 2 / 4 / 8 K slices by as many workgroups each, combined inside the launch (arrival ticket, write-through partials, per-slice
    flags, sum in slice order by the last arrival).  Every output against the whole-tile form; a row / column sample that covers the split tiles
    against the fp64 oracle; bit-reproducible over repeated launches (the sum does not depend on who arrives last); ticket / flag words zero and
    the sticky error word clear afterwards."""
    _check_balanced_tail(4, 128, dtype, act, K, N, M, slices)


@pytest.mark.parametrize("bits,gs,dtype,act,K,N,M", [
    (8, 32, torch.float16, False, 1024, 2304, 3800), (8, 128, torch.bfloat16, True, 2048, 2304, 7424), (3, 32, torch.float16, True, 2048, 2304, 3800),
    (3, 128, torch.bfloat16, False, 1024, 2304, 7424), (2, 64, torch.float16, False, 2048, 2304, 3800), (2, 32, torch.bfloat16, True, 1024, 2304, 7300),
    (4, 32, torch.float16, True, 2048, 2304, 7424), (4, 32, torch.bfloat16, False, 1024, 2304, 3800)])
def test_tiled_gemm_balanced_tail_other_packings(bits, gs, dtype, act, K, N, M):
    """The same on the 32-deep K-step forms of the kernel: 2- / 3- / 8-bit layers and 4-bit layers with 32-wide groups."""
    _check_balanced_tail(bits, gs, dtype, act, K, N, M)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("M,K,N,gs,act", [(9, 512, 96, 128, False), (16, 1024, 256, 32, False), (17, 1024, 256, 64, True),
                                          (32, 2048, 512, 128, False), (50, 4096, 1024, 128, True), (64, 224, 64, 32, False),
                                          (33, 11008, 512, 128, False)])
def test_strip16_batched_decode_kernel(M, K, N, gs, act, dtype):
    """8 < M <= 64, 4-bit: the 16-column-strip kernel (tuning.reserved[2] = 3; the default up to M = 16) against the
    fp64 oracle, against the 64-column skinny kernel (reserved[2] = 1), with one-hot rows (exact dequantised rows come back:
    catches any row/column/k-slot mix-up of the 16x16x32 fragment layouts), with a forced K split, and bit-reproducible."""
    L = O.random_quant_layer(K, N, 4, gs, act_order=act, seed=M + K + N, bias=True, dtype=dtype)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, gs, zero_mode="wrap")
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
    y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP)
    t3, t1, t3k = _tuning(path=3), _tuning(path=3), _tuning(path=3, ksplit=3)
    t3.reserved[_lib.LAB.GEMM_KERNEL], t1.reserved[_lib.LAB.GEMM_KERNEL], t3k.reserved[_lib.LAB.GEMM_KERNEL] = _lib.LAB.GEMM_STRIP16, _lib.LAB.GEMM_SKINNY, _lib.LAB.GEMM_STRIP16
    with torch.no_grad():
        y3, y3b = q(x.to(DEV), tuning=t3), q(x.to(DEV), tuning=t3)
        y1 = q(x.to(DEV), tuning=t1)
        y3k = q(x.to(DEV), tuning=t3k)
        y_auto = q(x.to(DEV))
    assert torch.equal(y3, y3b)
    if M <= 16:
        assert torch.equal(y_auto, y3), "4-bit, 8 < M <= 16: auto dispatch must take the 16-column-strip kernel"
    _assert_close(y3, y64, y64, dtype, K, "strip16 vs f64")
    _assert_close(y3k, y64, y64, dtype, K, "strip16 split-K vs f64")
    _assert_close(y1, y64, y64, dtype, K, "skinny64 vs f64")
    ks = (torch.arange(M) * 37 + 5) % K
    xo = torch.zeros(M, K, dtype=dtype)
    xo[torch.arange(M), ks] = 1.0
    with torch.no_grad():
        yo = q(xo.to(DEV), tuning=t3).cpu()
    W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 4, O.ZERO_WRAP)
    expect = (W[ks].float() + L["bias"].float()).to(dtype)
    assert torch.equal(yo, expect)


def test_gemm_matches_gemv_paths():
    """The three kernel families (fp32-math GEMV, matrix-core GEMV, MFMA GEMM) are three summation orders of the same
    exactly-dequantised products."""
    K, N, M = 1024, 512, 8
    L = O.random_quant_layer(K, N, 4, 128, seed=21)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(2)) - 0.5).half()
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, 4, 128)
    y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, 4, O.ZERO_WRAP)
    with torch.no_grad():
        ys = [q(x.to(DEV), tuning=_tuning(path=p)) for p in (1, 3, 5)]
    for y in ys:
        _assert_close(y, y64, y64, torch.float16, K, "path agreement")


@pytest.mark.parametrize("K,N,M,act", [(4096, 4096, 2048, True), (4096, 11008, 512, True), (11008, 4096, 300, False),
                                       (4096, 4096, 4096, False)])
def test_full_size_prefill_properties(K, N, M, act):
    """Llama-7B shapes at prefill sizes (BASELINE config 3 / north_star M=4096): (a) a row subset against
    the fp64 oracle on a column slice, (b) every output against x @ W with W = this library's own
    dequantize() (itself bit-exact vs the reference, tested above) multiplied in fp32 on the GPU,
    (c) row-block independence: rows [r0, r1) pushed alone give bit-identical outputs when the tile
    decomposition is unchanged, (d) bit reproducibility."""
    L = O.random_quant_layer(K, N, 4, 128, act_order=act, seed=K // 11 + N + M)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, 4, 128)
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(M, K, generator=gen) - 0.5).half().to(DEV)
    with torch.no_grad():
        y, yb = q(x), q(x)
    assert torch.equal(y, yb)
    W = q.dequantize().float()
    ref = x.float() @ W
    scale = float(ref.abs().max())
    err = float((y.float() - ref).abs().max())
    assert err <= 2e-3 * scale, (err, scale)          # fp16 output rounding (2^-11) + accumulation order
    # (a) fp64 oracle on 8 rows x 256 columns
    mode = O.reference_zero_mode(act, 4)
    assert q.resolved_zero_mode() == int(mode == O.ZERO_NOWRAP)
    n0 = (N // 3) // 32 * 32
    sl = slice(n0, n0 + 256)
    rows = torch.tensor([0, 1, 31, 32, M // 2, M - 33, M - 2, M - 1])
    y64 = O.forward_f64(x[rows].cpu(), L["qweight"][:, sl], L["qzeros"][:, n0 // 8:(n0 + 256) // 8], L["scales"][:, sl],
                        L["g_idx"] if act else None, None, 4, mode)
    _assert_close(y[rows][:, sl], y64, y64, torch.float16, K, "prefill rows vs f64")
    # (c) a 128-aligned row block on its own
    r0 = (M // 2) // 128 * 128
    t = _tuning(path=3, ksplit=1)
    t.reserved[_lib.LAB.GEMM_VARIANT] = _lib.LAB.VARIANT_ONE_K_GROUP                  # same K decomposition for both launches (no K groups inside the workgroup)
    with torch.no_grad():
        y_ns = q(x, tuning=t)
        yblk = q(x[r0:r0 + 128].contiguous(), tuning=t)
    assert torch.equal(yblk, y_ns[r0:r0 + 128])
    assert float((y_ns.float() - ref).abs().max()) <= 2e-3 * scale


@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("gs,K,N,M", [(128, 1024, 1024, 1), (128, 2048, 512, 3), (32, 512, 768, 2), (64, 1024, 96, 5),
                                     (128, 4096, 256, 8), (1024, 1024, 256, 4), (128, 11008, 64, 1)])
def test_matrix_core_gemv_bf16(gs, K, N, M, act):
    """4-bit bf16 through the matrix-core GEMV: B = 128 + w (bf16 magic number), sum_k x_k from a ones MFMA, fp32 fix-up."""
    L = O.random_quant_layer(K, N, 4, gs, dtype=torch.bfloat16, act_order=act, seed=K + N + M, bias=True)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(9)) - 0.5).bfloat16()
    for zm, mode in (("wrap", O.ZERO_WRAP), ("nowrap", O.ZERO_NOWRAP)):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, gs, zero_mode=zm)
        y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, mode)
        yref = O.forward(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, mode)
        with torch.no_grad():
            y = q(x.to(DEV), tuning=_tuning(path=5))
        _assert_close(y, yref, y64, torch.bfloat16, K, f"bf16 mfma zero={zm}")
        _assert_close(y, y64, y64, torch.bfloat16, K, f"bf16 mfma zero={zm} vs f64")
        # fp32 sums: the only bf16 rounding is the output's -> error vs fp64 within one bf16 ulp of the result scale
        assert float((y.double().cpu() - y64).abs().max()) <= 2.0 ** -8 * max(1.0, float(y64.abs().max()))


@pytest.mark.parametrize("gs,K,N,M", [(128, 1024, 1024, 1), (128, 2048, 512, 3), (32, 512, 768, 2), (64, 1024, 96, 5),
                                     (128, 4096, 256, 8), (1024, 1024, 256, 4), (128, 11008, 64, 1)])
@pytest.mark.parametrize("path", [5])
def test_direct_gemv_vs_oracle(gs, K, N, M, path):
    """The no-LDS-staging q4/fp16 GEMV (tuning.path = 5: matrix core; the v_dot2 variant, path = 4, was retired in round 6): x / scales /
    zeros read straight from L2."""
    L = O.random_quant_layer(K, N, 4, gs, seed=K + N + M, bias=True)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(9)) - 0.5).half()
    for zm, mode in (("wrap", O.ZERO_WRAP), ("nowrap", O.ZERO_NOWRAP)):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, gs, zero_mode=zm)
        y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, mode)
        yref = O.forward(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, mode)
        with torch.no_grad():
            y = q(x.to(DEV), tuning=_tuning(path=path))
        _assert_close(y, yref, y64, torch.float16, K, f"direct zero={zm}")
        _assert_close(y, y64, y64, torch.float16, K, f"direct zero={zm} vs f64")


# ---------------------------------------------------------------------------------- fused callers
@pytest.mark.parametrize("M", [1, 3, 8, 20, 130])
def test_fused_qkv_matches_separate_layers(M):
    """q/k/v concatenated along out_features (the reference's fused attention layout) = the three layers one by one."""
    from autogptq_amd.fused import fuse_qkv
    K = 1024
    Ls = [O.random_quant_layer(K, N, 4, 128, seed=40 + i, bias=True) for i, N in enumerate((512, 256, 256))]
    mods = [_module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, 128) for L in Ls]
    fused = fuse_qkv(*mods).to(DEV)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(3)) - 0.5).half()
    with torch.no_grad():
        y = fused(x.to(DEV))
        ys = [m(x.to(DEV)) for m in mods]
    assert tuple(y.shape) == (M, 1024)
    y64 = torch.cat([O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP) for L in Ls], dim=1)
    _assert_close(y, y64, y64, torch.float16, K, "fused qkv vs f64")
    _assert_close(y, torch.cat(ys, dim=1), y64, torch.float16, K, "fused qkv vs separate")


@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("M,dtype,bits", [(1, torch.float16, 4), (2, torch.float16, 4), (4, torch.float16, 4), (7, torch.float16, 4),
                                         (8, torch.float16, 4), (16, torch.float16, 4), (130, torch.float16, 4),
                                         (1, torch.bfloat16, 4), (3, torch.float16, 8), (40, torch.bfloat16, 3)])
def test_fused_gate_up_silu_mul(M, dtype, bits, act):
    """silu(x @ W_gate) * (x @ W_up) from ONE layer ([gate | up] columns, epilogue='silu_mul'): the fused matrix-core GEMV
    epilogue for 4-bit fp16 M <= 8, the staged y + elementwise pass otherwise.  Oracle: fp64 of the same expression."""
    from autogptq_amd.fused import fuse_gate_up
    K, N = 1024, 704                                  # 704 = 11 * 64: ragged against the 16-column strips' XCD remap
    Lg = O.random_quant_layer(K, N, bits, 128, dtype=dtype, seed=60, bias=True, act_order=act)
    Lu = O.random_quant_layer(K, N, bits, 128, dtype=dtype, seed=61, bias=True, act_order=act)
    Lu["g_idx"] = Lg["g_idx"].clone()                 # a fused pair shares its input permutation
    for L in (Lg, Lu):
        L["scales"] = (L["scales"].float() * 4).to(dtype)     # gate pre-activations of order 1: silu is exercised off its linear part
    mg = _module_from(Lg["qweight"], Lg["qzeros"], Lg["scales"], Lg["g_idx"], Lg["bias"], bits, 128)
    mu = _module_from(Lu["qweight"], Lu["qzeros"], Lu["scales"], Lu["g_idx"], Lu["bias"], bits, 128)
    fused = fuse_gate_up(mg, mu).to(DEV)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(4)) - 0.5).to(dtype)
    with torch.no_grad():
        y = fused(x.to(DEV))
        y2 = fused(x.to(DEV))
    assert tuple(y.shape) == (M, N) and torch.equal(y, y2)
    mode = O.reference_zero_mode(act, bits)
    g64 = O.forward_f64(x, Lg["qweight"], Lg["qzeros"], Lg["scales"], Lg["g_idx"], Lg["bias"], bits, mode)
    u64 = O.forward_f64(x, Lu["qweight"], Lu["qzeros"], Lu["scales"], Lu["g_idx"], Lu["bias"], bits, mode)
    ref = torch.nn.functional.silu(g64) * u64
    assert float(g64.abs().max()) > 1.0
    rtol = {torch.float16: 4e-3, torch.bfloat16: 3e-2}[dtype]       # up to 3 roundings to T on the unfused path
    scale = float(ref.abs().max())
    err = (y.double().cpu() - ref).abs()
    assert bool((err <= rtol * (ref.abs() + 0.05 * scale)).all()), float(err.max())


@pytest.mark.parametrize("M", [2, 3, 4, 5, 8])
@pytest.mark.parametrize("act", [False, True])
def test_fused_gate_up_wide_layer_small_batch(M, act):
    """A [gate | up] layer wide enough for the streamed 64-column-strip kernel (2 x 5632 columns = 176 strips): fused GEMV epilogue
    up to 2 rows, streamed kernel + elementwise pass from 3 rows (below that kernel's own threshold of 5 rows: the library asks
    for it through an internal tuning) -- same oracle, plan pinned."""
    from autogptq_amd.fused import fuse_gate_up
    K, N = 1024, 5632
    Lg = O.random_quant_layer(K, N, 4, 128, seed=160, bias=True, act_order=act)
    Lu = O.random_quant_layer(K, N, 4, 128, seed=161, bias=True, act_order=act)
    Lu["g_idx"] = Lg["g_idx"].clone()
    for L in (Lg, Lu):
        L["scales"] = (L["scales"].float() * 4).half()
    mg = _module_from(Lg["qweight"], Lg["qzeros"], Lg["scales"], Lg["g_idx"], Lg["bias"], 4, 128)
    mu = _module_from(Lu["qweight"], Lu["qzeros"], Lu["scales"], Lu["g_idx"], Lu["bias"], 4, 128)
    fused = fuse_gate_up(mg, mu).to(DEV)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half()
    with torch.no_grad():
        y = fused(x.to(DEV))
        y2 = fused(x.to(DEV))
    q = next(m for m in fused.modules() if isinstance(m, QuantLinear))
    d = _lib.describe_plan(q._layer, M)
    if M <= 2:      # (this file runs with QuantLinear.TILED_DECODE = False: the kernels of rounds 1-3; the pair form of the decode-copy kernel: test_gpu_tiled.py)
        assert d["epilogue"] == "fused", d
    else:
        assert (d["kernel"], d["epilogue"]) == ("stream64", "separate"), d
    assert tuple(y.shape) == (M, N) and torch.equal(y, y2)
    mode = O.reference_zero_mode(act, 4)
    g64 = O.forward_f64(x, Lg["qweight"], Lg["qzeros"], Lg["scales"], Lg["g_idx"], Lg["bias"], 4, mode)
    u64 = O.forward_f64(x, Lu["qweight"], Lu["qzeros"], Lu["scales"], Lu["g_idx"], Lu["bias"], 4, mode)
    ref = torch.nn.functional.silu(g64) * u64
    scale = float(ref.abs().max())
    err = (y.double().cpu() - ref).abs()
    assert bool((err <= 4e-3 * (ref.abs() + 0.05 * scale)).all()), float(err.max())


@pytest.mark.parametrize("M_first,M_then", [(16, 4), (8, 3), (32, 5)])
def test_unfused_epilogue_inner_k_split_uses_the_real_ticket_header(M_first, M_then):
    """Regression (round-2 advisor finding): a [gate | up] layer whose inner (epilogue-stripped) call splits K inside the launch.
    The inner call must take its arrival tickets from the workspace's real zeroed header, not from a header carved out of the
    body behind the staged y -- that region holds whatever an earlier, larger call left there (here: made certain by filling the
    whole body with 0xFF), so its tickets never reached ksplit - 1, the combine never ran and silu_mul saw stale y."""
    from autogptq_amd.fused import fuse_gate_up
    from autogptq_amd.qlinear_mi355x import reserve_workspace
    K, N = 4096, 4096                                  # [gate | up] = 8192 columns = 128 strips of 64: the planner splits K
    Lg = O.random_quant_layer(K, N, 4, 128, seed=260, bias=True)
    Lu = O.random_quant_layer(K, N, 4, 128, seed=261, bias=True)
    for L in (Lg, Lu):
        L["scales"] = (L["scales"].float() * 2).half()
    mg = _module_from(Lg["qweight"], Lg["qzeros"], Lg["scales"], Lg["g_idx"], Lg["bias"], 4, 128)
    mu = _module_from(Lu["qweight"], Lu["qzeros"], Lu["scales"], Lu["g_idx"], Lu["bias"], 4, 128)
    fused = fuse_gate_up(mg, mu).to(DEV)
    q = next(m for m in fused.modules() if isinstance(m, QuantLinear))
    q.post_init()
    plans = {M: _lib.describe_plan(q._layer, M) for M in (M_first, M_then)}
    assert any(int(d.get("ksplit", 1)) > 1 and d.get("epilogue") == "separate" for d in plans.values()), plans
    mode = O.reference_zero_mode(False, 4)
    for M in (M_first, M_then, M_first):
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half()
        with torch.no_grad():
            fused(x.to(DEV))                          # sizes the workspace for this M
            buf = reserve_workspace(torch.device(DEV), 1)
            torch.cuda.synchronize()
            buf[65536:].fill_(0xFF)                   # everything behind the ticket header is garbage before the call
            y = fused(x.to(DEV))
        g64 = O.forward_f64(x, Lg["qweight"], Lg["qzeros"], Lg["scales"], Lg["g_idx"], Lg["bias"], 4, mode)
        u64 = O.forward_f64(x, Lu["qweight"], Lu["qzeros"], Lu["scales"], Lu["g_idx"], Lu["bias"], 4, mode)
        ref = torch.nn.functional.silu(g64) * u64
        scale = float(ref.abs().max())
        err = (y.double().cpu() - ref).abs()
        # (y = [gate | up] is staged in fp16 on this path: a large negative gate value amplifies its rounding ~ |g| x through silu)
        assert bool((err <= 8e-3 * (ref.abs() + 0.05 * scale)).all()), (M, float(err.max()), scale, plans)
        assert int(buf[:32768].count_nonzero()) == 0, "the ticket half of the header must be left zero by every launch"


def _mlp_layers(K, I, N, bits, gs, dtype, act, seed):
    Lg = O.random_quant_layer(K, I, bits, gs, dtype=dtype, seed=seed, bias=True, act_order=act)
    Lu = O.random_quant_layer(K, I, bits, gs, dtype=dtype, seed=seed + 1, bias=True, act_order=act)
    Ld = O.random_quant_layer(I, N, bits, gs, dtype=dtype, seed=seed + 2, bias=True, act_order=act)
    Lg["scales"] = (Lg["scales"].float() * 4).to(dtype)             # gate pre-activations of order 1: silu off its linear part
    Lu["scales"] = (Lu["scales"].float() * 2).to(dtype)
    mods = [_module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], bits, gs) for L in (Lg, Lu, Ld)]
    return (Lg, Lu, Ld), mods


def _mlp_oracle(x, Ls, bits, act, dtype):
    """fp64 of the same function, with the activation rounded to the layer dtype where the library rounds it (fused_llama_mlp.py:237-242)."""
    mode = O.reference_zero_mode(act, bits)
    Lg, Lu, Ld = Ls
    g64 = O.forward_f64(x, Lg["qweight"], Lg["qzeros"], Lg["scales"], Lg["g_idx"], Lg["bias"], bits, mode)
    u64 = O.forward_f64(x, Lu["qweight"], Lu["qzeros"], Lu["scales"], Lu["g_idx"], Lu["bias"], bits, mode)
    a = (torch.nn.functional.silu(g64) * u64).to(dtype)
    return O.forward_f64(a, Ld["qweight"], Ld["qzeros"], Ld["scales"], Ld["g_idx"], Ld["bias"], bits, mode), float(g64.abs().max())


@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("M,dtype,bits", [(1, torch.float16, 4), (3, torch.float16, 4), (16, torch.float16, 4), (130, torch.float16, 4),
                                         (1, torch.bfloat16, 4), (2, torch.float16, 3), (40, torch.float16, 8)])
def test_mlp_forward_one_call(M, dtype, bits, act):
    """gptq_mlp_forward: down(silu(gate(x)) * up(x)) as ONE C-ABI call over three checkpoint layers (no concatenated copies), every packing,
    act-order per projection, any row count -- against the fp64 oracle of the same expression; bit-reproducible; checkpoint tensors untouched."""
    from autogptq_amd.qlinear_mi355x import mlp_forward
    K, I, N = 1024, 1408, 768
    Ls, (mg, mu, md) = _mlp_layers(K, I, N, bits, 128 if bits == 4 else 32, dtype, act, 400)
    before = [m.qweight.clone() for m in (mg, mu, md)]
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
    with torch.no_grad():
        y = mlp_forward(mg, mu, md, x.to(DEV))
        y2 = mlp_forward(mg, mu, md, x.to(DEV))
    assert tuple(y.shape) == (M, N) and y.dtype == dtype and torch.equal(y, y2)
    ref, gmax = _mlp_oracle(x, Ls, bits, act, dtype)
    assert gmax > 0.5
    # three roundings to T (gate, up, activation) in front of a K = I matmul: rtol on the value, atol on the output scale
    rtol, atol = {torch.float16: (4e-3, 2e-3), torch.bfloat16: (3e-2, 1.6e-2)}[dtype]
    scale = float(ref.abs().max())
    err = (y.double().cpu() - ref).abs()
    assert bool((err <= rtol * ref.abs() + atol * scale).all()), (float(err.max()), scale)
    for m, b in zip((mg, mu, md), before):
        assert torch.equal(m.qweight, b)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_mlp_forward_act_order_down_reads_the_permuted_activations_of_the_silu_pass(dtype):
    """Round 6: where the act-order `down` of gptq_mlp_forward runs a decode-copy GEMM (x permuted in the natural order of its re-sequenced rows), SiLU * mul and
    that permute are ONE pass (csrc/mlp.hip silu_mul2_permute_rows4_kernel) -- no permute launch of down's own, one round trip of the [M, I] activations less.
    The reference's fused MLP hands c_proj the activations in the order its rows are stored in (fused_llama_mlp.py:131-306; exllama's x_map,
    exllama/cuda_func/q4_matmul.cu:92-126).  Checked: the plan says so at the batched / prefill row counts and not at decode rows (in-kernel gather); the permuted
    activations left in the workspace are silu(g) * u of the STAGED gate / up outputs gathered through down's perm, entry by entry; the output against the fp64
    oracle of the whole expression; bit-reproducible; checkpoint tensors untouched."""
    from autogptq_amd import qlinear_mi355x as qm
    from autogptq_amd.qlinear_mi355x import mlp_forward
    K, I, N = 1024, 2048, 1024
    Ls, (mg, mu, md) = _mlp_layers(K, I, N, 4, 128, dtype, True, 640)
    for m in (mg, mu, md):
        m.post_init(tiled=True)                                         # (this file runs with QuantLinear.TILED_DECODE = False: these layers carry their decode copy)
    before = [m.qweight.clone() for m in (mg, mu, md)]
    esz = 2
    for M in (2, 8, 64, 300, 2048):
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
        with torch.no_grad():
            y = mlp_forward(mg, mu, md, x.to(DEV))
            y2 = mlp_forward(mg, mu, md, x.to(DEV))
        torch.cuda.synchronize()
        assert torch.equal(y, y2)
        plan = _lib.describe_mlp_plan(mg._layer, mu._layer, md._layer, M)
        pd = _lib.describe_plan(md._layer, M)
        fused = pd["path"] == "gemm" and pd["kernel"] in ("rows", "panel", "wide_sk", "wide_copy")      # the decode-copy GEMMs: x permuted in natural order
        assert (plan["down_permute"] == "fused") == fused, (M, plan, pd)
        assert fused == (M > 4), (M, pd)                                # (decode rows: the decode kernel gathers x itself)
        if fused:
            ws = qm._WORKSPACE[(torch.cuda.current_device(), int(torch.cuda.current_stream().cuda_stream))][0]
            sb = (M * I * esz + 255) // 256 * 256
            o = _lib.WS_HEADER_BYTES
            g = ws[o:o + M * I * esz].view(dtype).reshape(M, I).float()
            u = ws[o + sb:o + sb + M * I * esz].view(dtype).reshape(M, I).float()
            xp = ws[o + 2 * sb:o + 2 * sb + M * I * esz].view(dtype).reshape(M, I)
            want = (g / (1 + torch.exp(-g)) * u)[:, md._keepalive[6].long()]
            ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
            assert bool(((xp.float() - want).abs() <= 1.01 * ulp * want.abs() + 1e-7).all()), f"M={M}: the permuted activations are not silu(g) * u gathered through down's perm"
        ref, gmax = _mlp_oracle(x, Ls, 4, True, dtype)
        assert gmax > 0.5
        rtol, atol = {torch.float16: (4e-3, 2e-3), torch.bfloat16: (3e-2, 1.6e-2)}[dtype]
        scale = float(ref.abs().max())
        err = (y.double().cpu() - ref).abs()
        assert bool((err <= rtol * ref.abs() + atol * scale).all()), (M, float(err.max()), scale)
    for m, b in zip((mg, mu, md), before):
        assert torch.equal(m.qweight, b)


def test_mlp_forward_takes_no_path_override():
    """Round 3's one-launch persistent MLP kernel (tuning.path = 7) was a measured negative result and is a lab now (tools/lab/mlp_ring.hip): the
    product entry point refuses a path override loudly instead of silently running something else, and every product entry point is bit-reproducible
    again."""
    from autogptq_amd.qlinear_mi355x import mlp_forward, exchange_error
    Ls, (mg, mu, md) = _mlp_layers(1024, 2816, 1024, 4, 128, torch.float16, False, 501)
    x = (torch.rand(1, 1024, generator=torch.Generator().manual_seed(1)) - 0.5).half().to(DEV)
    with pytest.raises(_lib.GptqError, match="lab"):
        mlp_forward(mg, mu, md, x, tuning=_tuning(path=7))
    assert _lib.describe_mlp_plan(mg._layer, mu._layer, md._layer, 1)["kernel"] == "unfused"
    with torch.no_grad():
        y, y2 = mlp_forward(mg, mu, md, x), mlp_forward(mg, mu, md, x)
    assert torch.equal(y, y2)
    assert not exchange_error(DEV)


# ------------------------------------------------------------------------- callers around the path
def test_model_level_flow_make_quant_pack_post_init_forward():
    """make_quant -> pack_model (device pack) -> autogptq_post_init -> forward on a toy module: the quantized model tracks the
    float model whose weights were replaced by their dequantised values."""
    from autogptq_amd.model_utils import autogptq_post_init, pack_model

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.up = torch.nn.Linear(256, 512, bias=True)
            self.down = torch.nn.Linear(512, 256, bias=False)

        def forward(self, x):
            return self.down(torch.nn.functional.relu(self.up(x)))

    torch.manual_seed(1)
    m = Toy().half()
    quantizers, Wq = {}, {}
    for nme, lin in (("up", m.up), ("down", m.down)):
        s, z = O.minmax_quantize(lin.weight.data.float(), 4, 64)
        gi = torch.from_numpy(O.default_g_idx(lin.in_features, 64))
        quantizers[nme] = (None, s.half(), z.half(), gi)
        qw, qz, sc = O.pack(lin.weight.data.clone(), s.half(), z.half(), gi, 4, torch.float16)
        Wq[nme] = O.dequantize(qw, qz, sc, gi, 4, O.ZERO_WRAP).t().contiguous()      # [N, K]
    ref = Toy().half()
    ref.load_state_dict(m.state_dict())
    ref.up.weight.data, ref.down.weight.data = Wq["up"], Wq["down"]
    pack_model(m, quantizers, 4, 64)
    m = autogptq_post_init(m.to(DEV), use_act_order=False, max_input_length=64)
    for M in (1, 5, 40):
        x = (torch.rand(M, 256, generator=torch.Generator().manual_seed(M)) - 0.5).half()
        with torch.no_grad():
            y = m(x.to(DEV)).float().cpu()
            yr = ref.float()(x.float())
        assert float((y - yr).abs().max()) <= 4e-3 * max(1.0, float(yr.abs().max())), M


def test_row_and_column_shards_on_one_gpu():
    """Both tensor-parallel splits, all shards evaluated on one GPU: column shards concatenate to the full output, row
    shards sum to it (what the RCCL all-gather / all-reduce assemble across ranks)."""
    from autogptq_amd.tensor_parallel import ColumnParallelQuantLinear, RowParallelQuantLinear
    K, N, T = 2048, 1024, 4
    L = O.random_quant_layer(K, N, 4, 128, seed=33, bias=True)
    full = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, 128)
    x = (torch.rand(3, K, generator=torch.Generator().manual_seed(8)) - 0.5).half().to(DEV)
    y64 = O.forward_f64(x.cpu(), L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP)
    with torch.no_grad():
        cols = [ColumnParallelQuantLinear.from_full(full, r, T, device=DEV, gather_output=False)(x) for r in range(T)]
        _assert_close(torch.cat(cols, dim=-1), y64, y64, torch.float16, K, "column shards")
        rows = [RowParallelQuantLinear.from_full(full, r, T, device=DEV, input_is_parallel=False) for r in range(T)]
        part = sum(rp.local(x[:, rp.k0:rp.k1].contiguous()).float() for rp in rows) + L["bias"].float().to(DEV)
        _assert_close(part.half(), y64, y64, torch.float16, K, "row shards")


# ------------------------------------------------------------------- AWQ ingest (auto_gptq/modeling/_utils.py:525-701)
AWQ_GOLDEN = ["awq_k128_n64_g32.npz", "awq_k256_n128_g128.npz", "awq_k64_n64_g32_zero_edges.npz"]
_GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("fname", AWQ_GOLDEN)
def test_awq_ingest_matches_reference_outputs(fname):
    """gptq_awq_unpack / gptq_pack_* / gptq_awq_repack against what the reference's unpack_awq and pack_from_tensors
    themselves returned for the same AWQ words (tests/golden/make_golden_awq.py): every array bit for bit."""
    from autogptq_amd import awq
    d = np.load(os.path.join(_GOLDEN_DIR, fname))
    gs = int(d["group_size"])
    aq, az, sc = (torch.from_numpy(d[k]) for k in ("awq_qweight", "awq_qzeros", "scales"))
    W, Z = awq.unpack_awq(aq, az, sc, 4, gs)
    assert W.dtype == torch.float16 and Z.dtype == torch.int8 and tuple(W.shape) == (int(d["N"]), int(d["K"]))
    assert np.array_equal(W.contiguous().cpu().numpy().view(np.uint16), d["fp16_weight"].view(np.uint16))
    assert np.array_equal(Z.cpu().numpy(), d["zeros"])
    qw, qz = awq.pack_from_tensors(W, Z, sc, 4, gs)
    assert np.array_equal(qw.cpu().numpy(), d["qweight"]) and np.array_equal(qz.cpu().numpy(), d["qzeros"])
    qw2, qz2 = awq.repack_awq_to_gptq(aq, az, gs)
    assert np.array_equal(qw2.cpu().numpy(), d["qweight"]) and np.array_equal(qz2.cpu().numpy(), d["qzeros"])
    zt = torch.from_numpy(np.ascontiguousarray(d["z"].astype(np.int8).T))         # [N, G]: the function transposes first
    from oracle import awq_oracle as A
    assert np.array_equal(awq.awq_reverse_reorder_int_tensor(zt.to(DEV), 4).cpu().numpy(), A.reverse_reorder(zt.numpy()))


@pytest.mark.parametrize("K,N,gs", [(1024, 512, 128), (4096, 4096, 128), (11008, 4096, 128), (512, 2056, 64), (64, 8, 8)])
def test_awq_ingest_random_vs_oracle(K, N, gs):
    """Llama-7B sizes and ragged widths (N/8 not a multiple of the block): the integer repack against the numpy oracle on
    every word; the fp16 unpack on a row sample; and the round trip unpack -> pack_from_tensors == repack (a property that
    needs no oracle, checked at full size)."""
    from autogptq_amd import awq
    from oracle import awq_oracle as A
    rng = np.random.default_rng(K + N)
    G = K // gs
    aq = rng.integers(-2**31, 2**31 - 1, size=(K, N // 8), dtype=np.int64).astype(np.int32)
    az = rng.integers(-2**31, 2**31 - 1, size=(G, N // 8), dtype=np.int64).astype(np.int32)
    sc = (0.002 * (1 + rng.random((G, N)))).astype(np.float16)
    qw, qz = awq.repack_awq_to_gptq(torch.from_numpy(aq), torch.from_numpy(az), gs)
    eqw, eqz = A.awq_to_gptq(aq, az)
    assert np.array_equal(qw.cpu().numpy(), eqw) and np.array_equal(qz.cpu().numpy(), eqz)
    W, Z = awq.unpack_awq(torch.from_numpy(aq), torch.from_numpy(az), torch.from_numpy(sc), 4, gs)
    rows = np.unique(np.concatenate([np.arange(0, K, max(1, K // 97)), [K - 1]]))
    Wo, Zo = A.unpack_awq(aq[rows], az[rows // gs], sc[rows // gs], 1)          # row k with its own group row: group_size 1
    assert np.array_equal(W.T[torch.from_numpy(rows).to(DEV)].cpu().numpy().view(np.uint16), Wo.T.view(np.uint16))
    assert np.array_equal(Z.cpu().numpy()[rows // gs], Zo)
    if K % 32 == 0 and N % 32 == 0:        # pack() geometry of the reference (qlinear_cuda.py:39-40)
        qw3, qz3 = awq.pack_from_tensors(W, Z, torch.from_numpy(sc), 4, gs)
        assert torch.equal(qw3, qw) and torch.equal(qz3, qz)


def test_awq_ingested_layer_forward():
    """End to end: AWQ words -> gptq_awq_repack -> QuantLinear (cuda_old zero convention, the one that maps the stored
    (z - 1) & 15 back to z) -> forward == x @ (s * (w - z)) of the AWQ checkpoint, including z = 0 and z = 15."""
    from autogptq_amd import awq
    from oracle import awq_oracle as A
    rng = np.random.default_rng(9)
    K, N, gs = 1024, 768, 128
    w = rng.integers(0, 16, size=(K, N))
    z = rng.integers(0, 16, size=(K // gs, N))
    z[0, :16], z[1, :16] = 0, 15
    s = torch.from_numpy((0.002 * (1 + rng.random((K // gs, N)))).astype(np.float16))
    qw, qz = awq.repack_awq_to_gptq(torch.from_numpy(A.awq_pack(w)), torch.from_numpy(A.awq_pack(z)), gs)
    q = _module_from(qw.cpu(), qz.cpu(), s, None, None, 4, gs, zero_mode="wrap")
    Wd = torch.from_numpy(w - np.repeat(z, gs, axis=0)).double() * s.double().repeat_interleave(gs, 0)
    assert torch.equal(q.dequantize().cpu(), Wd.to(torch.float16))
    for M in (1, 4, 48, 300):
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half()
        y64 = x.double() @ Wd
        with torch.no_grad():
            y = q(x.to(DEV))
        _assert_close(y, y64, y64, torch.float16, K, f"AWQ-ingested layer, M={M}")


@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("M", [1, 3, 12, 40, 200])
def test_longest_k_of_the_baseline_configs(M, act):
    """K = 28672 (Llama-2-70B down_proj, BASELINE config 4): the longest reduction any config has -- beyond the K range of
    the act-order matrix-core GEMV (x row staged in LDS, K <= 24576), so the dispatcher has to fall back correctly."""
    K, N = 28672, 256
    L = O.random_quant_layer(K, N, 4, 128, act_order=act, seed=M, bias=True)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, 128, zero_mode="wrap")
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half()
    y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP)
    with torch.no_grad():
        y, yb = q(x.to(DEV)), q(x.to(DEV))
    assert torch.equal(y, yb)
    _assert_close(y, y64, y64, torch.float16, K, f"K=28672 M={M} act={act}")


# ------------------------------------------------------------------- the reference's backend-vs-cuda_old grid
# tests/test_hpu_linear.py:102-181 compares a backend with the cuda_old Python path over group sizes, square layer sizes, 13
# (scales, weight, zeros) value patterns and two dtypes; the same grid here, with the oracle standing in for cuda_old.
HPU_PATTERNS = [("normal", "normal", "normal"), ("normal", "normal", "range"), ("normal", "normal", "zeros"), ("ones", "zeros", "zeros"),
                ("ones", "zeros", "eights"), ("ones", "range", "zeros"), ("ones", "range", "ones"), ("ones", "7", "ones"),
                ("ones", "zeros", "range"), ("ones", "zeros", "ones"), ("ones", "range", "range"), ("range", "range", "range"),
                ("range", "range", "zeros")]


def _pattern_layer(K, N, gs, pattern, dtype, bias, seed):
    """(linear, scales [N, G] fp32, zeros [N, G] int32) in the value patterns of the reference grid (:121-150)."""
    sv, wv, zv = pattern
    gen = torch.Generator().manual_seed(seed)
    G = K // gs
    W = torch.randn(N, K, generator=gen) * 0.05
    s = W.reshape(N, G, gs).abs().amax(dim=2) / 7 + 1e-4                      # symmetric 4-bit scale per (column, group)
    if sv == "ones":
        s = torch.ones_like(s)
    elif sv == "range":
        s = torch.arange(1, s.numel() + 1, dtype=torch.float32).reshape(G, N).t().contiguous()
    if wv == "normal":
        Wq = (torch.clamp(torch.round(W / s.repeat_interleave(gs, 1)), -8, 7)) * s.repeat_interleave(gs, 1)
    elif wv == "zeros":
        Wq = torch.zeros(N, K)
    elif wv == "range":
        Wq = torch.arange(8, dtype=torch.float32).repeat(N * K // 8).reshape(N, K)
    else:
        Wq = torch.full((N, K), float(wv))
    if zv == "zeros":
        z = torch.zeros(N, G, dtype=torch.int32)
    elif zv == "range":
        z = torch.arange(1, 9, dtype=torch.int32).repeat(N * G // 8).reshape(G, N).t().contiguous()
    elif zv == "eights":
        z = torch.full((N, G), 8, dtype=torch.int32)
    elif zv == "ones":
        z = torch.ones(N, G, dtype=torch.int32)
    else:                                   # "normal": the symmetric mid-point
        z = torch.full((N, G), 8, dtype=torch.int32)
    lin = torch.nn.Linear(K, N, bias=bias)
    lin.weight.data = Wq.to(dtype)
    if bias:
        lin.bias.data = (torch.randn(N, generator=gen) * 0.1).to(dtype)
    return lin, s, z


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("pi", range(len(HPU_PATTERNS)), ids=["-".join(p) for p in HPU_PATTERNS])
@pytest.mark.parametrize("KN", [64, 128, 512])
@pytest.mark.parametrize("gs", [16, 32, 128])
def test_reference_backend_grid(gs, KN, pi, dtype):
    if KN < gs:
        pytest.skip("infeatures < group_size (the reference grid skips these too, test_hpu_linear.py:115-116)")
    K = N = KN
    bias = bool((pi + KN // 64) & 1)
    lin, s, z = _pattern_layer(K, N, gs, HPU_PATTERNS[pi], dtype, bias, seed=pi * 7 + gs)
    q = QuantLinear(4, gs, K, N, bias, weight_dtype=dtype)
    q.pack(lin, s.clone(), z.clone(), g_idx=None)                 # device pack (gptq_pack_weights / gptq_pack_zeros)
    qw, qz, sc = O.pack(lin.weight.data.clone(), s.clone(), z.clone(), None, 4, dtype)
    assert torch.equal(q.qweight.cpu(), qw) and torch.equal(q.qzeros.cpu(), qz) and torch.equal(q.scales.cpu(), sc)
    q = q.to(DEV)
    b = lin.bias.data.clone() if bias else None
    for M in (1, 5, 64):
        x = torch.rand(M, K, generator=torch.Generator().manual_seed(M)).to(dtype)
        yref = O.forward(x, qw, qz, sc, None, b, 4, O.ZERO_WRAP)
        y64 = O.forward_f64(x, qw, qz, sc, None, b, 4, O.ZERO_WRAP)
        with torch.no_grad():
            y = q(x.to(DEV))
        assert y.dtype == dtype
        _assert_close(y, yref, y64, dtype, K, f"{HPU_PATTERNS[pi]} gs={gs} M={M} vs reference order")
        _assert_close(y, y64, y64, dtype, K, f"{HPU_PATTERNS[pi]} gs={gs} M={M} vs f64")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K,N,gs", [(15360, 64, 128), (8192, 32, 64), (4096, 4096, 128), (14336, 256, 128), (24576, 64, 128), (1024, 96, 32)])
def test_act_order_decode_long_k_small_n(K, N, gs, dtype):
    """act-order, M = 1: the whole x row is staged in LDS (perm[] points anywhere in [0, K)).  Tiny N forces a deep K split,
    i.e. small workgroups that still have to stage the WHOLE row; the default plan and forced splits of the matrix-core GEMV."""
    L = O.random_quant_layer(K, N, 4, gs, act_order=True, seed=K + N, bias=True, dtype=dtype)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, gs, zero_mode="nowrap")
    x = (torch.rand(1, K, generator=torch.Generator().manual_seed(K)) - 0.5).to(dtype)
    y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_NOWRAP)
    with torch.no_grad():
        y = q(x.to(DEV))
        yb = q(x.to(DEV))
    assert torch.equal(y, yb)
    _assert_close(y, y64, y64, dtype, K, "act-order decode, default plan")
    for ks in (2, 16, 32):
        with torch.no_grad():
            yk = q(x.to(DEV), tuning=_tuning(path=5, ksplit=ks))
        _assert_close(yk, y64, y64, dtype, K, f"act-order decode, ksplit={ks}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K,N,M", [(512, 11008, 1), (1024, 5120, 2), (256, 12288, 4), (512, 13824, 3), (1024, 10240, 1), (512, 11008, 5)])
def test_wide_layers_take_wide_strips(K, N, M, dtype):
    """Plain layers wider than 4096 columns: the planner picks 64- or 32-column strips (>= 160 workgroups) for M <= 4 -- fp16 and
    bf16 -- and the result must not depend on the strip width (explicit 16-column strips give the same sums in another order)."""
    L = O.random_quant_layer(K, N, 4, 128, seed=K + N + M, bias=True, dtype=dtype)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, 128)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
    y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP)
    with torch.no_grad():
        y, yb = q(x.to(DEV)), q(x.to(DEV))
        y4 = q(x.to(DEV), tuning=_tuning(path=5, lanes_n=4))
        # 64-column strips exist where the planner itself can choose them: fp16, up to 4 rows (round 6); bf16 / 5+ rows are refused, not replaced
        if dtype == torch.float16 and M <= 4:
            y16 = q(x.to(DEV), tuning=_tuning(path=5, lanes_n=16))
        else:
            y16 = y4
            if dtype == torch.float16:
                with pytest.raises(_lib.GptqError, match="lanes_n"):
                    q(x.to(DEV), tuning=_tuning(path=5, lanes_n=16))
    assert torch.equal(y, yb)
    for t, what in ((y, "auto"), (y4, "16-column strips"), (y16, "64-column strips")):
        _assert_close(t, y64, y64, dtype, K, what)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("K,N,M", [(5120, 5120, 1), (5120, 5120, 3), (6656, 5152, 2), (5120, 16384, 4), (13824, 5120, 1)])
def test_large_model_layers_take_the_streamed_gemv(K, N, M, dtype):
    """K and N >= 5120 (Llama-13B / 33B / 70B projections), M <= 4: the default plan is the streamed (LDS-DMA) GEMV with 32-column
    strips (64 from 16384 columns), 8 waves x 4 rows per lane -- against the fp64 oracle on two column slices (one at the ragged
    right edge: N = 5152 is not a multiple of the strip), against the register kernel, twice."""
    L = O.random_quant_layer(K, N, 4, 128, seed=K // 3 + N + M, bias=True, dtype=dtype)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], None, L["bias"], 4, 128)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
    with torch.no_grad():
        y, yb = q(x.to(DEV)), q(x.to(DEV))
        yr = q(x.to(DEV), tuning=_tuning(path=5))
    d = _lib.describe_plan(q._layer, M)
    if M >= 3 and N >= 10240:                   # 160+ strips of 64 columns and 3..4 rows: the batched-decode kernel, unsplit
        assert (d["kernel"], d["ksplit"]) == ("stream64", 1), d
    else:
        assert (d["kernel"], d["ln"], d["waves"], d["u"]) == ("stream", 16 if N >= 16384 else 8, 8, 4), d
    assert torch.equal(y, yb)
    mode = O.reference_zero_mode(False, 4)
    for n0 in ((N // 2) // 32 * 32, N - 96):
        sl = slice(n0, n0 + 96)
        y64 = O.forward_f64(x, L["qweight"][:, sl], L["qzeros"][:, n0 // 8:(n0 + 96) // 8], L["scales"][:, sl], None, L["bias"][sl], 4, mode)
        _assert_close(y[:, sl], y64, y64, dtype, K, f"streamed default, columns {n0}:{n0 + 96}")
        _assert_close(yr[:, sl], y64, y64, dtype, K, "register kernel")


@pytest.mark.parametrize("fname", ["marlin_k256_n256_g128.npz", "marlin_k512_n512_g128.npz"])       # (the single-group file: CPU tests)
def test_marlin_checkpoint_layer_forward(fname):
    """A layer serialised in the reference's Marlin format (tests/golden/marlin_*.npz = the reference's own pack()) converted
    to GPTQ tensors by autogptq_amd.marlin and run through this backend: its dequantised matrix is exactly the fake-quantised
    weight that was packed, and forward = x @ that weight (+ bias) at GEMV, strip and tiled sizes."""
    from autogptq_amd import marlin
    d = np.load(os.path.join(_GOLDEN_DIR, fname))
    gs = int(d["group_size"])
    qw, qz, sc = marlin.marlin_to_gptq(torch.from_numpy(d["B"]), torch.from_numpy(d["s"]), gs)
    bias = torch.from_numpy(d["bias"]) if d["bias"].size else None
    q = _module_from(qw, qz, sc, None, bias, 4, gs)
    Wq = torch.from_numpy(d["Wq"]).t().contiguous()                     # [K, N] fp16
    assert torch.equal(q.dequantize().cpu(), Wq)
    K = int(d["K"])
    for M in (1, 4, 12, 40, 200):
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half()
        y64 = x.double() @ Wq.double()
        if bias is not None:
            y64 = y64 + bias.double()
        with torch.no_grad():
            y = q(x.to(DEV))
        _assert_close(y, y64, y64, torch.float16, K, f"Marlin-format layer, M={M}")


# ------------------------------------------------------------------------- scratch / device hygiene
def test_workspace_outgrown_after_capture_stays_valid_for_the_graph():
    """A hipGraph captured with a small scratch keeps working after a later call needed (and got) a much larger one: outgrown
    buffers are retired, never freed (the captured launches have their address baked in)."""
    from autogptq_amd import qlinear_mi355x as QM
    L = O.random_quant_layer(8192, 1024, 4, 128, seed=11)                 # narrow layer: K split -> needs scratch at M = 1
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], None, None, 4, 128)
    assert _lib.describe_plan(q._layer if q._layer else q.post_init()._layer, 1)["ksplit"] > 1
    x = (torch.rand(1, 8192) - 0.5).half().to(DEV)
    with torch.no_grad():
        ref = q(x).clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        y = q(x)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    retired_before = len(QM._RETIRED)
    La = O.random_quant_layer(4096, 4096, 4, 128, act_order=True, seed=12)   # act-order prefill: permuted x + slabs, tens of MB
    qa = _module_from(La["qweight"], La["qzeros"], La["scales"], La["g_idx"], None, 4, 128)
    xb = (torch.rand(2048, 4096) - 0.5).half().to(DEV)
    side = torch.cuda.Stream()
    for st in (torch.cuda.current_stream(), side):
        with torch.cuda.stream(st), torch.no_grad():
            qa(xb)
            junk = [torch.randn(1 << 20, device=DEV) for _ in range(8)]       # would land on freed scratch memory, if any were freed
    torch.cuda.synchronize()
    del junk
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(y, ref)
    assert len(QM._RETIRED) >= retired_before
    keys = [k for k in QM._WORKSPACE if k[0] == torch.cuda.current_device()]
    assert len(keys) >= 2, "scratch is keyed by (device, stream)"


def test_pack_on_gpu_module_from_cpu_quantizer_outputs_and_device_mismatch_error():
    """pack() leaves EVERY buffer (incl. g_idx and bias) on the module's device; a module whose buffers were split across
    devices by hand gets a Python error from post_init instead of a GPU fault."""
    lin = torch.nn.Linear(256, 128, bias=True).half()
    s, z = O.minmax_quantize(lin.weight.data.float(), 4, 64)
    gi = torch.from_numpy(O.default_g_idx(256, 64))[torch.randperm(256, generator=torch.Generator().manual_seed(0))].contiguous()
    q = QuantLinear(4, 64, 256, 128, True).to(DEV)
    q.pack(lin, s.half(), z.half(), gi)                       # CPU tensors into a GPU module
    for name in ("qweight", "qzeros", "scales", "g_idx", "bias"):
        assert getattr(q, name).device.type == "cuda", name
    x = (torch.rand(3, 256) - 0.5).half()
    qw, qz, sc = O.pack(lin.weight.data.clone(), s.half(), z.half(), gi, 4, torch.float16)
    ref = O.forward_f64(x, qw, qz, sc, gi, lin.bias.data, 4, O.ZERO_NOWRAP)
    with torch.no_grad():
        y = q(x.to(DEV))
    _assert_close(y, ref, ref, torch.float16, 256, "pack on GPU module")
    q2 = QuantLinear(4, 64, 256, 128, True).to(DEV)
    q2.g_idx = q2.g_idx.cpu()
    with pytest.raises(RuntimeError, match="g_idx is on cpu"):
        q2.post_init()
    bad = QuantLinear(4, 64, 256, 128, False).to(DEV)
    bad.g_idx = (torch.arange(256, dtype=torch.int32) % 5).to(DEV)         # group 4 does not exist (G = 4)
    with pytest.raises(_lib.GptqError, match="outside"):
        bad.post_init()


# ------------------------------------------------------------------------- streamed GEMV (LDS DMA) + gptq_forward_multi
def _stream_tuning(ln, waves, u, ksplit):
    t = _tuning(path=6, lanes_n=ln, waves=waves, ksplit=ksplit)
    t.reserved[_lib.LAB.DEPTH] = u
    return t


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("ln,waves,u,ksplit", [(4, 4, 2, 1), (4, 8, 4, 1), (4, 16, 8, 1), (16, 8, 4, 2), (16, 16, 8, 3), (4, 8, 2, 4), (16, 4, 2, 1), (8, 8, 2, 1), (8, 4, 4, 2)])
@pytest.mark.parametrize("K,N,gs,M", [(1024, 512, 128, 1), (2048, 96, 64, 3), (4096, 1056, 128, 4), (512, 2048, 128, 2)])
def test_streamed_gemv_vs_oracle(K, N, gs, M, ln, waves, u, ksplit, dtype):
    """gemv_q4_stream_kernel (tuning.path = 6): every launch geometry, incl. ragged last strips (N = 96, 1056 with 64-column
    strips), several passes over K (small waves x u), the in-launch K-split combine, bias, both zero conventions."""
    if u > gs // 8:
        pytest.skip("u rows of a lane must lie in one group")
    L = O.random_quant_layer(K, N, 4, gs, dtype=dtype, seed=K + N + M, bias=True)
    for zm in ("wrap", "nowrap"):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], None, L["bias"], 4, gs, zero_mode=zm)
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
        t = _stream_tuning(ln, waves, u, ksplit)
        q.post_init()
        d = _lib.describe_plan(q._layer, M, t)
        assert d["kernel"] == "stream" and d["ln"] == ln and d["waves"] == waves and d["u"] == u, d
        with torch.no_grad():
            y1 = q(x.to(DEV), tuning=t)
            y2 = q(x.to(DEV), tuning=t)                       # second launch: tickets were reset by the first
        assert torch.equal(y1, y2)
        mode = O.ZERO_WRAP if zm == "wrap" else O.ZERO_NOWRAP
        ref = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], None, L["bias"], 4, mode)
        _assert_close(y1, ref, ref, dtype, K, f"streamed gemv ln={ln} waves={waves} u={u} ksplit={ksplit}")


@pytest.mark.parametrize("bits,ln,waves,u,ksplit", [(8, 4, 16, 4, 1), (8, 4, 4, 2, 1), (8, 8, 8, 4, 2), (8, 4, 8, 4, 3), (8, 8, 2, 4, 1),
                                                    (3, 4, 16, 1, 1), (3, 4, 4, 1, 2), (3, 8, 8, 1, 1), (3, 4, 8, 2, 1), (3, 8, 4, 2, 4),
                                                    (2, 4, 16, 2, 1), (2, 8, 4, 4, 2), (2, 4, 8, 4, 1), (2, 8, 8, 2, 3)])      # (8 units per lane: lab-only forms, retired in round 6)
@pytest.mark.parametrize("K,N,gs,M", [(1024, 512, 32, 1), (2048, 96, 64, 3), (4096, 1056, 128, 4), (512, 2048, 32, 2), (11008, 256, 32, 1)])
def test_streamed_gemv_3_and_8_bit(K, N, gs, M, bits, ln, waves, u, ksplit):
    """gemv_qx_stream_kernel (tuning.path = 6 on 3- / 8-bit fp16 layers): the packing units (one word of 4 values / three words of 32) by LDS DMA,
    the packed magic-number decode, every launch geometry incl. ragged last strips, several passes over K, the in-launch K-split combine, bias,
    both zero conventions (3-bit zero-points straddle words); against the fp64 oracle and BIT-equal to the register kernel with the same geometry
    where that exists."""
    kpu = {3: 32, 8: 4, 2: 16}[bits]
    if u > gs // kpu or (K // kpu) % u:
        pytest.skip("u units of a lane must lie in one group")
    L = O.random_quant_layer(K, N, bits, gs, dtype=torch.float16, seed=K + N + M + bits, bias=True)
    for zm in ("wrap", "nowrap"):
        q = _module_from(L["qweight"], L["qzeros"], L["scales"], None, L["bias"], bits, gs, zero_mode=zm)
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half()
        t = _stream_tuning(ln, waves, u, ksplit)
        q.post_init()
        d = _lib.describe_plan(q._layer, M, t)
        assert d["kernel"] == "stream" and d["ln"] == ln and d["waves"] == waves and d["u"] == u, d
        with torch.no_grad():
            y1 = q(x.to(DEV), tuning=t)
            y2 = q(x.to(DEV), tuning=t)
        assert torch.equal(y1, y2)
        mode = O.ZERO_WRAP if zm == "wrap" else O.ZERO_NOWRAP
        ref = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], None, L["bias"], bits, mode)
        _assert_close(y1, ref, ref, torch.float16, K, f"streamed {bits}-bit gemv ln={ln} waves={waves} u={u} ksplit={ksplit}")
        # one-hot rows on the bias-free layer: the exact dequantised rows come back (the scale multiplies the fp32 group sum, one rounding)
        q0 = _module_from(L["qweight"], L["qzeros"], L["scales"], None, None, bits, gs, zero_mode=zm)
        ks_ = (torch.arange(M) * 131 + 7) % K
        xo = torch.zeros(M, K, dtype=torch.float16)
        xo[torch.arange(M), ks_] = 1.0
        with torch.no_grad():
            yo = q0(xo.to(DEV), tuning=t).cpu()
        W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], None, bits, mode)
        assert torch.equal(yo, W[ks_])


@pytest.mark.parametrize("bits,gs", [(8, 32), (3, 32), (8, 128), (3, 64), (2, 64), (2, 32)])
@pytest.mark.parametrize("M", [1, 3])
def test_forward_multi_3_and_8_bit_one_launch(bits, gs, M):
    """gptq_forward_multi on 3- / 8-bit fp16 layers that share x: one gemv_qx_stream_kernel launch (forced and by default), every layer against
    the fp64 oracle and equal to its own single-layer forward within the tolerance."""
    from autogptq_amd.qlinear_mi355x import forward_multi
    K, widths = 2048, (512, 96, 1024)
    Ls = [O.random_quant_layer(K, n, bits, gs, seed=31 + i + M + bits, bias=(i == 1), dtype=torch.float16) for i, n in enumerate(widths)]
    qs = [_module_from(L["qweight"], L["qzeros"], L["scales"], None, L["bias"], bits, gs) for L in Ls]
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half().to(DEV)
    t = _tuning(path=6)
    with torch.no_grad():
        yf = forward_multi(qs, x, tuning=t)
        yd = forward_multi(qs, x)
        sep = [q(x) for q in qs]
    mode = O.reference_zero_mode(False, bits)
    for y, d, s_, L in zip(yf, yd, sep, Ls):
        ref = O.forward_f64(x.cpu(), L["qweight"], L["qzeros"], L["scales"], None, L["bias"], bits, mode)
        _assert_close(y, ref, ref, torch.float16, K, "qx multi (forced) vs oracle")
        _assert_close(d, ref, ref, torch.float16, K, "qx multi (default) vs oracle")
        _assert_close(s_, ref, ref, torch.float16, K, "single layer vs oracle")


@pytest.mark.parametrize("M", [1, 2, 4, 5, 40])
@pytest.mark.parametrize("act", [False, True])
def test_forward_multi_equals_layer_by_layer(M, act):
    """gptq_forward_multi: q/k/v-like (3 layers, different widths) and gate/up-like (2 layers) groups give exactly the
    outputs of separate forward calls -- one streamed launch for M <= 4 plain layers, layer by layer otherwise (act-order,
    M > 4)."""
    from autogptq_amd.qlinear_mi355x import forward_multi
    K = 2048
    Ls = [O.random_quant_layer(K, n, 4, 128, act_order=act, seed=31 + i, bias=(i == 1)) for i, n in enumerate((512, 160, 1024))]
    qs = [_module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"] if act else None, L["bias"], 4, 128) for L in Ls]
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half().to(DEV)
    with torch.no_grad():
        ys = forward_multi(qs, x)
        ys2 = forward_multi(qs, x)
        sep = [q(x) for q in qs]
    mode = O.reference_zero_mode(act, 4)
    for y, y2, s, L in zip(ys, ys2, sep, Ls):
        assert torch.equal(y, y2)
        ref = O.forward_f64(x.cpu(), L["qweight"], L["qzeros"], L["scales"], L["g_idx"] if act else None, L["bias"], 4, mode)
        _assert_close(y, ref, ref, torch.float16, K, "forward_multi vs oracle")
        _assert_close(y, s, ref, torch.float16, K, "forward_multi vs separate")
    if M <= 4 and not act:                       # the single-layer streamed kernel (its own K split: same values within rounding)
        t = _tuning(path=6)
        with torch.no_grad():
            one = [q(x, tuning=t) for q in qs]
        for y, o, L in zip(ys, one, Ls):
            _assert_close(o, y, y.double(), torch.float16, K, "single-layer streamed vs multi")
    # inside a hipGraph, with a K split (narrow layers): tickets survive capture + replays
    narrow = [O.random_quant_layer(4096, n, 4, 128, seed=77 + n) for n in (128, 256)]
    qn = [_module_from(L["qweight"], L["qzeros"], L["scales"], None, None, 4, 128) for L in narrow]
    xn = (torch.rand(1, 4096) - 0.5).half().to(DEV)
    with torch.no_grad():
        eager = forward_multi(qn, xn)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        cap = forward_multi(qn, xn)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    for a, b in zip(eager, cap):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M", [64, 200, 2048])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_forward_multi_shares_one_permuted_x_between_layers_of_one_act_order(M, dtype):
    """q / k / v (and gate / up) of a GPTQ checkpoint carry ONE g_idx (the order is the argsort of the Hessian diagonal of their common input; the
    reference's fused q/k/v caller hands the kernels q_proj's order for all three, fused_llama_attn.py:188).  forward_multi points such layers at one
    `perm` buffer (share_act_order) and the C ABI then permutes x ONCE per call: the outputs are BIT-IDENTICAL to separate calls and agree with the fp64
    oracle; layers whose g_idx differ keep their own perms (and their own permute launches)."""
    from autogptq_amd.qlinear_mi355x import forward_multi, share_act_order
    K = 2048
    base = O.random_quant_layer(K, 512, 4, 128, act_order=True, dtype=dtype, seed=5)
    Ls = [base] + [O.random_quant_layer(K, n, 4, 128, act_order=True, dtype=dtype, seed=6 + i) for i, n in enumerate((256, 1024))]
    for L in Ls[1:]:
        L["g_idx"] = base["g_idx"].clone()                       # the same activation order (the packed rows are random words: any order is a valid layer)
    qs = [_module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, 4, 128) for L in Ls]
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype).to(DEV)
    with torch.no_grad():
        sep = [q(x) for q in qs]                                 # before sharing: each layer its own perm and permute launch
        ys = forward_multi(qs, x)
        ys2 = forward_multi(qs, x)
    assert qs[1]._layer.perm == qs[0]._layer.perm and qs[2]._layer.perm == qs[0]._layer.perm
    mode = O.reference_zero_mode(True, 4)
    for y, y2, s_, L in zip(ys, ys2, sep, Ls):
        assert torch.equal(y, y2) and torch.equal(y, s_)
        ref = O.forward_f64(x.cpu(), L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, 4, mode)
        _assert_close(y, ref, ref, dtype, K, "shared permuted x vs oracle")
    with torch.no_grad():
        for q, s_ in zip(qs, sep):                               # a sharer on its own still works (its perm pointer is layer 0's buffer)
            assert torch.equal(q(x), s_)
    other = O.random_quant_layer(K, 256, 4, 128, act_order=True, dtype=dtype, seed=99)
    qo = _module_from(other["qweight"], other["qzeros"], other["scales"], other["g_idx"], None, 4, 128)
    qo.post_init()
    assert not share_act_order([qs[0], qo])
    with torch.no_grad():
        ym = forward_multi([qs[0], qo], x)
        assert torch.equal(ym[0], sep[0]) and torch.equal(ym[1], qo(x))


# ------------------------------------------------------------------------- fp32 I/O above M = 8: exact-f32 matrix core
@pytest.mark.parametrize("act", [False, True])
@pytest.mark.parametrize("bits,gs,K,N,M", [(4, 128, 1024, 512, 9), (4, 32, 512, 160, 40), (3, 32, 1024, 256, 130), (8, 32, 512, 384, 300),
                                           (2, 16, 256, 96, 64), (4, 128, 4096, 4096, 2048), (8, 128, 2048, 1056, 77)])
def test_fp32_gemm_vs_oracle(bits, gs, K, N, M, act):
    """fp32 layers (scales / x / out fp32: the reference's use_cuda_fp16=False and act-order native paths, qlinear_cuda_old.py:
    251-290, qlinear_cuda.py:216-250) at M > 8 run gemm_f32_kernel: weights are the exact fp32 s*(w-z), accumulation is the
    f32 MFMA.  fp64 oracle with the fp32 tolerance, one-hot rows exact, bias, ragged N (160, 1056) and M, act-order."""
    L = O.random_quant_layer(K, N, bits, gs, act_order=act, dtype=torch.float32, seed=bits + K + N + M, bias=True)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"] if act else None, L["bias"], bits, gs)
    q.post_init()
    few_tiles = M <= 64 or -(-M // 128) * -(-N // 128) < 64          # the planner keeps such launches on the GEMV (want_gemm, fp32 branch)
    assert _lib.describe_plan(q._layer, M)["kernel"] == ("generic" if few_tiles else "f32_mfma")
    t_gemm = _tuning(path=3)
    gen = torch.Generator().manual_seed(M)
    x = (torch.rand(M, K, generator=gen) - 0.5).float()
    hot = [0, K - 1, gs, 10]
    for r, k in enumerate(hot):
        x[r].zero_()
        x[r, k] = 1.0
    with torch.no_grad():
        y = q(x.to(DEV), tuning=t_gemm)               # the matrix-core kernel, whatever the planner would pick
        yb = q(x.to(DEV), tuning=t_gemm)
        W = q.dequantize()
    assert y.dtype == torch.float32 and torch.equal(y, yb)
    for r, k in enumerate(hot):
        assert torch.equal(y[r], W[k] + q.bias), (r, k)
    rows = torch.arange(0, M, max(1, M // 37))
    cols = slice(0, min(N, 512))
    mode = O.reference_zero_mode(act, bits)
    y64 = O.forward_f64(x[rows], L["qweight"][:, cols], L["qzeros"][:, : cols.stop * bits // 32], L["scales"][:, cols], L["g_idx"] if act else None,
                        L["bias"][cols], bits, mode)
    _assert_close(y[rows][:, cols], y64, y64, torch.float32, K, "fp32 gemm vs f64")
    # and the GEMV path it replaces gives the same values within the same tolerance (different summation order)
    with torch.no_grad():
        yg = q(x[:8].to(DEV))
    _assert_close(yg, y[:8].double(), y64, torch.float32, K, "fp32 gemv rows vs gemm rows")


# ------------------------------------------------------------------------- batched decode: 64-column strips by LDS DMA
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("waves,u,ksplit", [(0, 0, 0), (16, 8, 1), (8, 2, 3), (4, 4, 2), (2, 1, 5), (8, 4, 8)])
@pytest.mark.parametrize("M,K,N,gs,act", [(5, 512, 96, 128, False), (8, 4096, 1024, 128, True), (16, 1024, 256, 32, False),
                                          (17, 1024, 160, 64, True), (32, 2048, 512, 128, False), (50, 4096, 1056, 128, True),
                                          (64, 224, 64, 32, False), (33, 11008, 512, 128, False)])
def test_stream64_batched_decode_kernel(M, K, N, gs, act, dtype, waves, u, ksplit):
    """4 < M <= 64, 4-bit: gemm_stream64_kernel (tuning.reserved[2] = 4; the default in that range) against the fp64 oracle for
    default and forced launch geometries (waves x K-steps in flight x in-launch K split, ragged last strip: N = 96 / 160 / 1056,
    ragged row tile, K ranges that do not divide by waves x u, groups of one K-step), with one-hot rows (the exact dequantised
    rows come back: catches any row / column / k-slot mix-up of the fragment layouts), twice (tickets reset) and bit-reproducible."""
    L = O.random_quant_layer(K, N, 4, gs, act_order=act, seed=M + K + N, bias=True, dtype=dtype)
    q = _module_from(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, gs, zero_mode="wrap")
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
    y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"], 4, O.ZERO_WRAP)
    t = _tuning(path=3, waves=waves, ksplit=ksplit)
    t.reserved[_lib.LAB.DEPTH], t.reserved[_lib.LAB.GEMM_KERNEL] = u, _lib.LAB.GEMM_STREAM64
    q.post_init()
    d = _lib.describe_plan(q._layer, M, t)
    assert d["kernel"] == "stream64", d
    with torch.no_grad():
        y, yb = q(x.to(DEV), tuning=t), q(x.to(DEV), tuning=t)
    assert torch.equal(y, yb)
    _assert_close(y, y64, y64, dtype, K, f"stream64 {d} vs f64")
    ks = (torch.arange(M) * 37 + 5) % K
    xo = torch.zeros(M, K, dtype=dtype)
    xo[torch.arange(M), ks] = 1.0
    with torch.no_grad():
        yo = q(xo.to(DEV), tuning=t).cpu()
    W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 4, O.ZERO_WRAP)
    expect = (W[ks].float() + L["bias"].float()).to(dtype)
    assert torch.equal(yo, expect)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("waves,u,ksplit", [(0, 0, 0), (16, 4, 1), (8, 2, 3), (4, 2, 8)])
@pytest.mark.parametrize("M,K,widths,gs", [(5, 1024, (512, 160, 1024), 128), (16, 2048, (1056, 96), 64), (33, 4096, (256, 256, 256, 64), 128),
                                           (64, 512, (704, 704), 32)])
def test_stream64_multi_layer_launch(M, K, widths, gs, dtype, waves, u, ksplit):
    """gptq_forward_multi at 4 < M <= 64: 2..4 plain layers that share x in ONE gemm_stream64_kernel launch (forced with
    tuning.path = 3 / reserved[2] = 4, and by default where the planner prefers it) -- every layer against the fp64 oracle and
    against its own single-layer forward, ragged last strips inside the concatenation, bias on some layers, tickets reset."""
    from autogptq_amd.qlinear_mi355x import forward_multi
    Ls = [O.random_quant_layer(K, n, 4, gs, seed=77 + i + M, bias=(i % 2 == 1), dtype=dtype) for i, n in enumerate(widths)]
    qs = [_module_from(L["qweight"], L["qzeros"], L["scales"], None, L["bias"], 4, gs) for L in Ls]
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype).to(DEV)
    t = _tuning(path=3, waves=waves, ksplit=ksplit)
    t.reserved[_lib.LAB.DEPTH], t.reserved[_lib.LAB.GEMM_KERNEL] = u, _lib.LAB.GEMM_STREAM64
    with torch.no_grad():
        ys = forward_multi(qs, x, tuning=t)
        ys2 = forward_multi(qs, x, tuning=t)
        yd = forward_multi(qs, x)
        sep = [q(x) for q in qs]
    mode = O.reference_zero_mode(False, 4)
    for y, y2, d, s_, L in zip(ys, ys2, yd, sep, Ls):
        assert torch.equal(y, y2)
        ref = O.forward_f64(x.cpu(), L["qweight"], L["qzeros"], L["scales"], None, L["bias"], 4, mode)
        _assert_close(y, ref, ref, dtype, K, "stream64 multi vs oracle")
        _assert_close(d, ref, ref, dtype, K, "forward_multi default vs oracle")
        _assert_close(y, s_, ref, dtype, K, "stream64 multi vs separate")
