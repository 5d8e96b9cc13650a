"""CPU: pin the oracle (oracle/gptq_oracle.py) against the reference's own known-answer vectors
and against fixtures produced by the reference classes themselves (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import gptq_oracle as O


def _kat(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return z


@pytest.mark.parametrize("fname,zero_class", [
    ("kat_cuda_old_reference_1024.npz", False),   # tests/test_q4.py:1060-1122 / 1899-1941
    ("kat_cuda_old_reference_1024.npz", True),
    ("kat_reference_old_half_256.npz", False),    # tests/test_q4.py:1752-1802 (use_half2=True)
    ("kat_reference_old_no_half_256.npz", False), # (use_half2=False)
])
def test_known_answer_vectors(golden_dir, fname, zero_class):
    z = _kat(golden_dir, fname)
    k, n = int(z["k"]), int(z["n"])
    dtype = {"float16": torch.float16, "float32": torch.float32}[str(z["dtype"])]
    qweight, qzeros, scales, x = O.golden_recipe_inputs(k, n, dtype=dtype)
    mode = O.reference_zero_mode(zero_class, 4)
    y = O.forward(x, qweight, qzeros, scales, None, None, 4, mode)[0][0]
    ref = torch.from_numpy(z["y"]).to(dtype)
    assert torch.allclose(y, ref, rtol=float(z["rtol"]), atol=float(z["atol"])), (y - ref).abs().max()


def test_unpack_matches_reference_ints(ref_case):
    """qweight -> ints must reproduce pack()'s input ints for in-range values (round trip)."""
    c = ref_case
    w = O.unpack_weights(c.qweight, c.bits)
    assert w.shape == (c.K, c.N)
    assert w.max() <= 2 ** c.bits - 1
    # re-packing the unpacked fields reproduces the packed words bit-for-bit
    assert np.array_equal(O.pack_rows(w.astype(np.uint32), c.bits), c.qweight.numpy())
    zf = O.unpack_rows(np.ascontiguousarray(c.qzeros.numpy().T), c.bits).T
    assert np.array_equal(np.ascontiguousarray(O.pack_rows(np.ascontiguousarray(zf.astype(np.uint32).T), c.bits).T),
                          c.qzeros.numpy())


def test_pack_bit_exact(ref_case):
    c = ref_case
    W = c.W.to(c.dtype)
    qw, qz, sc = O.pack(W, c.scale.to(c.qparams_dtype), c.zero.to(c.qparams_dtype),
                        c.g_idx.numpy() if c.desc_act_class else None, c.bits, out_dtype=c.dtype)
    assert torch.equal(qw, c.qweight)
    assert torch.equal(qz, c.qzeros)
    assert torch.equal(sc, c.scales)


def test_dequant_bit_exact(ref_case):
    """Dequantised matrix == reference forward(identity): exact (one rounding per element)."""
    c = ref_case
    mode = O.reference_zero_mode(c.desc_act_class, c.bits)
    Wdq = O.dequantize(c.qweight, c.qzeros, c.scales, c.g_idx, c.bits, mode)
    assert Wdq.dtype == c.dtype
    # forward(I) rounds (I @ W) to dtype, which is exact for a one-hot row; bias subtraction may
    # cost an ulp in fp16, so compare exactly only without bias
    if c.bias is None:
        assert torch.equal(Wdq, c.Wdq)
    else:
        assert torch.allclose(Wdq.float(), c.Wdq.float(), rtol=2e-3, atol=2e-3)


def test_reference_speed_port_is_the_same_function(ref_case):
    """dequantize_torch / forward_fast (the broadcast shift + mask form bench.py's cpu_baseline times) are bit-identical to
    the field-by-field restatement on every fixture, with the fixture's g_idx and with the sequential-groups shortcut."""
    c = ref_case
    mode = O.reference_zero_mode(c.desc_act_class, c.bits)
    assert torch.equal(O.dequantize_torch(c.qweight, c.qzeros, c.scales, c.g_idx, c.bits, mode),
                       O.dequantize(c.qweight, c.qzeros, c.scales, c.g_idx, c.bits, mode))
    if not c.act_order:
        assert torch.equal(O.dequantize_torch(c.qweight, c.qzeros, c.scales, None, c.bits, mode),
                           O.dequantize(c.qweight, c.qzeros, c.scales, None, c.bits, mode))
    assert torch.equal(O.forward_fast(c.x, c.qweight, c.qzeros, c.scales, c.g_idx, c.bias, c.bits, mode),
                       O.forward(c.x, c.qweight, c.qzeros, c.scales, c.g_idx, c.bias, c.bits, mode))


def test_forward_matches_reference(ref_case):
    c = ref_case
    mode = O.reference_zero_mode(c.desc_act_class, c.bits)
    y = O.forward(c.x, c.qweight, c.qzeros, c.scales, c.g_idx, c.bias, c.bits, mode)
    tol = {torch.float32: (1e-5, 1e-6), torch.float16: (2e-3, 2e-3), torch.bfloat16: (2e-2, 2e-2)}[c.dtype]
    assert y.dtype == c.dtype and y.shape == c.y.shape
    assert torch.allclose(y.float(), c.y.float(), rtol=tol[0], atol=tol[1]), (y.float() - c.y.float()).abs().max()
    # and the exact-math variant agrees to the same tolerance
    y64 = O.forward_f64(c.x, c.qweight, c.qzeros, c.scales, c.g_idx, c.bias, c.bits, mode)
    assert torch.allclose(y64.float(), c.y.float(), rtol=tol[0] * 4, atol=tol[1] * 4)


def test_zero_mode_fork_is_visible():
    """all-ones stored zeros, q=0, scale=1, x=1, K=32, 4-bit:
    wrap -> 0.0, nowrap -> -512.0."""
    K, N, bits = 32, 32, 4
    qweight = torch.zeros((K // 8, N), dtype=torch.int32)
    qzeros = torch.full((1, N // 8), -1, dtype=torch.int32)
    scales = torch.ones((1, N), dtype=torch.float32)
    x = torch.ones((1, K), dtype=torch.float32)
    yw = O.forward(x, qweight, qzeros, scales, None, None, bits, O.ZERO_WRAP)
    yn = O.forward(x, qweight, qzeros, scales, None, None, bits, O.ZERO_NOWRAP)
    assert torch.all(yw == 0.0) and torch.all(yn == -512.0)


@pytest.mark.parametrize("bits", [2, 3, 4, 8])
def test_pack_unpack_roundtrip_random(bits):
    rng = np.random.default_rng(bits)
    vals = rng.integers(0, 2 ** bits, size=(96, 40), dtype=np.uint32)
    packed = O.pack_rows(vals, bits)
    assert packed.shape == (96 // 32 * bits, 40)
    assert np.array_equal(O.unpack_rows(packed, bits), vals.astype(np.uint16))


def test_sequential_permutation_is_stable_sort():
    g = np.array([2, 0, 1, 0, 2, 1, 0], dtype=np.int32)
    assert O.sequential_permutation(g).tolist() == [1, 3, 6, 2, 5, 0, 4]


# ---- the C restatement (oracle/gptq_oracle.c) agrees with the numpy one and with the reference
def test_c_oracle_matches(ref_case):
    from oracle import c_oracle as C
    c = ref_case
    mode = O.reference_zero_mode(c.desc_act_class, c.bits)
    assert np.array_equal(C.unpack_weights(c.qweight.numpy(), c.bits), O.unpack_weights(c.qweight, c.bits))
    assert np.array_equal(C.unpack_zeros(c.qzeros.numpy(), c.bits, mode == O.ZERO_NOWRAP),
                          O.unpack_zeros(c.qzeros, c.bits, mode))
    y = C.forward_f64(c.x.float().numpy(), c.qweight.numpy(), c.qzeros.numpy(), c.scales.float().numpy(),
                      c.g_idx.numpy(), None if c.bias is None else c.bias.float().numpy(),
                      c.bits, c.group_size, mode == O.ZERO_NOWRAP)
    y64 = O.forward_f64(c.x, c.qweight, c.qzeros, c.scales, c.g_idx, c.bias, c.bits, mode).numpy()
    assert np.allclose(y, y64, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("bits", [2, 3, 4, 8])
def test_c_oracle_random_words(bits):
    from oracle import c_oracle as C
    L = O.random_quant_layer(256, 128, bits, 64, seed=bits)
    assert np.array_equal(C.unpack_weights(L["qweight"].numpy(), bits), O.unpack_weights(L["qweight"], bits))
    for mode in (O.ZERO_WRAP, O.ZERO_NOWRAP):
        assert np.array_equal(C.unpack_zeros(L["qzeros"].numpy(), bits, mode == O.ZERO_NOWRAP),
                              O.unpack_zeros(L["qzeros"], bits, mode))


@pytest.mark.parametrize("bits,K,N,gs", [(4, 256, 128, 64), (4, 160, 64, 32), (3, 256, 64, 128), (3, 96, 32, 32), (8, 128, 64, 64), (8, 96, 32, 32),
                                         (2, 256, 64, 64), (2, 160, 32, 32)])
def test_c_oracle_decode_copy_matches_the_numpy_restatement(bits, K, N, gs):
    """The decode copy (include/gptq_mi355x.h, gptq_prepack_decode) stated twice: source-first with reshapes in gptq_oracle.py, destination-first bit by
    bit in gptq_oracle.c; ragged K (a last chunk that is part padding) included.  The device kernels are pinned to the numpy one in tests/test_gpu_tiled.py."""
    from oracle import c_oracle as C
    L = O.random_quant_layer(K, N, bits, gs, seed=17 * bits + K)
    want = O.decode_copy_weights(L["qweight"], bits).numpy().view(np.uint32)
    got = C.decode_copy_weights(L["qweight"].numpy(), bits)
    assert got.shape == want.shape and np.array_equal(got, want)
    sb = L["scales"].contiguous().view(torch.int16).numpy().view(np.uint16)
    for mode in (O.ZERO_WRAP, O.ZERO_NOWRAP):
        wantc = O.decode_copy_consts(L["qzeros"], L["scales"], mode, bits).numpy()
        gotc = C.decode_copy_consts(L["qzeros"].numpy(), sb, bits, mode == O.ZERO_NOWRAP)
        assert gotc.shape == wantc.shape and np.array_equal(gotc, wantc)


# ---------------------------------------------------------------- AWQ ingest (auto_gptq/modeling/_utils.py:525-701)
AWQ_GOLDEN = ["awq_k128_n64_g32.npz", "awq_k256_n128_g128.npz", "awq_k64_n64_g32_zero_edges.npz"]


@pytest.mark.parametrize("fname", AWQ_GOLDEN)
def test_awq_oracle_matches_reference_outputs(golden_dir, fname):
    """tests/golden/awq_*.npz hold what the reference's unpack_awq / pack_from_tensors returned (make_golden_awq.py);
    the numpy restatement must reproduce every array bit for bit, and the one-pass integer composition must agree."""
    from oracle import awq_oracle as A
    d = np.load(os.path.join(golden_dir, fname))
    gs = int(d["group_size"])
    assert np.array_equal(A.awq_pack(d["w"].astype(np.int64)), d["awq_qweight"])
    assert np.array_equal(A.awq_pack(d["z"].astype(np.int64)), d["awq_qzeros"])
    W, Z = A.unpack_awq(d["awq_qweight"], d["awq_qzeros"], d["scales"], gs)
    assert np.array_equal(W.view(np.uint16), d["fp16_weight"].view(np.uint16))
    assert Z.dtype == np.int8 and np.array_equal(Z, d["zeros"])
    qw, qz = A.pack_from_tensors(d["fp16_weight"], d["zeros"], d["scales"], gs)
    assert np.array_equal(qw, d["qweight"]) and np.array_equal(qz, d["qzeros"])
    qw2, qz2 = A.awq_to_gptq(d["awq_qweight"], d["awq_qzeros"])
    assert np.array_equal(qw2, d["qweight"]) and np.array_equal(qz2, d["qzeros"])


def test_awq_ingested_layer_dequantises_to_awq_weights():
    """GPTQ fields (z - 1) & 15 read back with the +1-then-mask convention (cuda_old) give z again, also for z = 0 and 15:
    the ingested layer's W equals AWQ's s * (w - z)."""
    from oracle import awq_oracle as A
    rng = np.random.default_rng(3)
    K, N, gs = 128, 64, 32
    w = rng.integers(0, 16, size=(K, N))
    z = rng.integers(0, 16, size=(K // gs, N))
    z[0, :4], z[1, :4] = 0, 15
    s = torch.from_numpy((0.002 * (1 + rng.random((K // gs, N)))).astype(np.float16))
    qw, qz = A.awq_to_gptq(A.awq_pack(w), A.awq_pack(z))
    W = O.dequantize(torch.from_numpy(qw), torch.from_numpy(qz), s, None, 4, O.ZERO_WRAP)
    expect = (torch.from_numpy(w - np.repeat(z, gs, axis=0)).to(torch.float16) * s.repeat_interleave(gs, 0))
    assert torch.equal(W, expect)


# ---------------------------------------------------------------- Marlin checkpoint format (qlinear_marlin.py:51-176)
MARLIN_GOLDEN = ["marlin_k128_n256_g128.npz", "marlin_k256_n256_g128.npz", "marlin_k512_n512_g128.npz"]


@pytest.mark.parametrize("fname", MARLIN_GOLDEN)
def test_marlin_oracle_matches_reference_pack(golden_dir, fname):
    """tests/golden/marlin_*.npz hold what the reference's QuantLinear.pack wrote (make_golden_marlin.py): the numpy
    restatement reproduces B and s bit for bit, its inverse recovers the integers and the natural scale order, and the GPTQ
    tensors derived from them dequantise exactly to the fake-quantised weight that was packed."""
    from oracle import marlin_oracle as M
    d = np.load(os.path.join(golden_dir, fname))
    gs = int(d["group_size"])
    B, sm = M.pack_ints(np.ascontiguousarray(d["ints"].T).astype(np.int64), np.ascontiguousarray(d["scales"].T), gs)
    assert np.array_equal(B, d["B"]) and np.array_equal(sm.view(np.uint16), d["s"].view(np.uint16))
    w, s = M.unpack_ints(d["B"], d["s"], gs)
    assert np.array_equal(w, d["ints"].T) and np.array_equal(s.view(np.uint16), np.ascontiguousarray(d["scales"].T).view(np.uint16))
    qw, qz, sc = M.to_gptq(d["B"], d["s"], gs)
    for mode in (O.ZERO_WRAP, O.ZERO_NOWRAP):
        W = O.dequantize(torch.from_numpy(qw), torch.from_numpy(qz), torch.from_numpy(sc), None, 4, mode)
        assert torch.equal(W, torch.from_numpy(d["Wq"]).t())
