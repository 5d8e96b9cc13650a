"""CPU oracle for the GPTQ QuantLinear hot path  --  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
``autogptq_amd`` (the product) never imports anything under ``oracle/``.

It restates, independently and in vectorised numpy / torch-CPU, the algorithm of
the reference's pure-PyTorch ``QuantLinear`` (the path that runs when the CUDA
extension is absent, ``BUILD_CUDA_EXT=0``).  All ``file:line`` citations are
relative to the AutoGPTQ reference tree (v0.8.0.dev0):

* bit layout / ``pack``            auto_gptq/nn_modules/qlinear/qlinear_cuda.py:108-203
                                   (identical body, sequential groups: qlinear_cuda_old.py:110-200)
* unpack, no act-order ("wrap")    auto_gptq/nn_modules/qlinear/qlinear_cuda_old.py:291-349
* unpack, act-order   ("nowrap")   auto_gptq/nn_modules/qlinear/qlinear_cuda.py:253-302
* matmul + epilogue                qlinear_cuda_old.py:350-355 / qlinear_cuda.py:313-317

Parity pinning: this oracle is checked (tests/test_oracle_golden.py) against
(1) the three known-answer vectors of the reference's own tests
    (tests/test_q4.py:29-1056, 1230-1489, 1491-1750), and
(2) outputs of the reference classes themselves, generated in the build
    container by tests/golden/make_golden.py and committed under tests/golden/.

Layout recap (K = in_features, N = out_features, P = 32 // bits):
  qweight int32 [K/32*bits, N]   word r of column n holds k = P*r .. P*r+P-1, LSB first
  qzeros  int32 [G, N/32*bits]   word c of group g holds (zero-1) for n = P*c .. P*c+P-1
  scales  fp    [G, N]
  g_idx   int32 [K]              group of input feature k
  3-bit: 32 values live in 3 consecutive words; values 10 and 21 straddle words.
"""
from __future__ import annotations

import numpy as np
import torch

ZERO_WRAP = "wrap"      # z = (field + 1) & maxq     qlinear_cuda_old.py:301-304  (2/4/8-bit only)
ZERO_NOWRAP = "nowrap"  # z = (field & maxq) + 1     qlinear_cuda.py:262-264, and every 3-bit path

SUPPORTED_BITS = (2, 3, 4, 8)


def reference_zero_mode(desc_act_class: bool, bits: int) -> str:
    """Which zero-point convention the reference class being replaced uses.

    ``cuda_old`` (no act-order) wraps for 2/4/8-bit (qlinear_cuda_old.py:301-304) but its
    3-bit branch adds 1 *after* the mask and never re-masks (qlinear_cuda_old.py:324-330);
    ``cuda`` (act-order) never wraps (qlinear_cuda.py:262-264, 279-285).
    """
    if desc_act_class or bits == 3:
        return ZERO_NOWRAP
    return ZERO_WRAP


# --------------------------------------------------------------------------- #
# integer field extraction (bit-exact target)
# --------------------------------------------------------------------------- #
def _bit_positions(bits: int):
    """(word_index, shift, nbits_in_this_word) pieces for each of the values stored in one
    32-bit-aligned unit: 1 word for 2/4/8-bit, 3 words (32 values) for 3-bit.

    3-bit unit (qlinear_cuda.py:144-162): word0 = v0..v9 @0..29, v10[1:0] @30..31;
    word1 = v10[2] @0, v11..v20 @1..30, v21[0] @31; word2 = v21[2:1] @0..1, v22..v31 @2..31.
    """
    if bits in (2, 4, 8):
        return [[(0, bits * j, bits)] for j in range(32 // bits)]
    assert bits == 3
    out = []
    for v in range(32):
        lo = 3 * v           # absolute bit offset inside the 96-bit unit
        hi = lo + 2
        if lo // 32 == hi // 32:
            out.append([(lo // 32, lo % 32, 3)])
        else:
            n_lo = 32 - (lo % 32)
            out.append([(lo // 32, lo % 32, n_lo), (hi // 32, 0, 3 - n_lo)])
    return out


def unpack_rows(q: np.ndarray, bits: int) -> np.ndarray:
    """Unpack along axis 0: int32 [R, C] -> uint16 [R*32/bits, C] raw fields in [0, 2**bits).

    Used for qweight (R = K/32*bits).  Logical (unsigned) shifts: the stored int32 is a
    reinterpreted uint32 (qlinear_cuda.py:133,166).
    """
    assert bits in SUPPORTED_BITS
    u = np.ascontiguousarray(q).view(np.uint32) if q.dtype == np.int32 else q.astype(np.uint32)
    words_per_unit = 1 if bits != 3 else 3
    vals_per_unit = 32 // bits if bits != 3 else 32
    assert u.shape[0] % words_per_unit == 0, "row count must be a multiple of the packing unit"
    units = u.reshape(u.shape[0] // words_per_unit, words_per_unit, *u.shape[1:])
    out = np.empty((units.shape[0], vals_per_unit) + u.shape[1:], dtype=np.uint16)
    for v, pieces in enumerate(_bit_positions(bits)):
        acc = np.zeros((units.shape[0],) + u.shape[1:], dtype=np.uint32)
        have = 0
        for (w, sh, nb) in pieces:
            acc |= ((units[:, w] >> np.uint32(sh)) & np.uint32((1 << nb) - 1)) << np.uint32(have)
            have += nb
        out[:, v] = acc.astype(np.uint16)
    return out.reshape((units.shape[0] * vals_per_unit,) + u.shape[1:])


def unpack_weights(qweight, bits: int) -> np.ndarray:
    """qweight int32 [K/32*bits, N] -> uint16 [K, N], w[k,n] in [0, maxq]."""
    q = qweight.cpu().numpy() if isinstance(qweight, torch.Tensor) else np.asarray(qweight)
    return unpack_rows(q, bits)


def unpack_zeros(qzeros, bits: int, zero_mode: str) -> np.ndarray:
    """qzeros int32 [G, N/32*bits] -> int32 [G, N] zero-points *as used in dequant*.

    wrap   : (field + 1) & maxq      -> [0, maxq]
    nowrap : field + 1               -> [1, maxq + 1]
    """
    q = qzeros.cpu().numpy() if isinstance(qzeros, torch.Tensor) else np.asarray(qzeros)
    fields = unpack_rows(np.ascontiguousarray(q.T), bits).T.astype(np.int32)  # [G, N]
    z = fields + 1
    if zero_mode == ZERO_WRAP:
        z &= (1 << bits) - 1
    elif zero_mode != ZERO_NOWRAP:
        raise ValueError(f"unknown zero_mode {zero_mode!r}")
    return np.ascontiguousarray(z)


# --------------------------------------------------------------------------- #
# dequant + matmul (floating point; tolerance-based target, dequant is exact)
# --------------------------------------------------------------------------- #
def default_g_idx(K: int, group_size: int) -> np.ndarray:
    return (np.arange(K, dtype=np.int64) // group_size).astype(np.int32)  # qlinear_cuda.py:72-75


def dequantize(qweight, qzeros, scales: torch.Tensor, g_idx, bits: int, zero_mode: str) -> torch.Tensor:
    """W[k,n] = scales[g(k),n] * (w[k,n] - z[g(k),n]) in the *scales dtype*.

    Mirrors the reference's arithmetic order: integer subtract first, then ONE multiply in the
    scales dtype (fp16 product is the correctly rounded exact product)
    (qlinear_cuda_old.py:348, qlinear_cuda.py:302).
    """
    w = unpack_weights(qweight, bits).astype(np.int32)            # [K, N]
    z = unpack_zeros(qzeros, bits, zero_mode)                     # [G, N]
    K = w.shape[0]
    if g_idx is None:
        G = z.shape[0]
        gs = -(-K // G)
        g = default_g_idx(K, gs)
    else:
        g = (g_idx.cpu().numpy() if isinstance(g_idx, torch.Tensor) else np.asarray(g_idx)).astype(np.int64)
    assert g.shape[0] == K, "len(g_idx) must equal in_features (fused-QKV g_idx is out of scope)"
    diff = torch.from_numpy((w - z[g]).astype(np.int16))          # exact, |diff| <= 256
    s = scales.cpu()[torch.from_numpy(g.astype(np.int64))]        # [K, N] gather
    return s * diff.to(s.dtype)                                   # int -> float is exact


def dequantize_torch(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, g_idx, bits: int, zero_mode: str) -> torch.Tensor:
    """Same result as dequantize() for 2/4/8-bit layers, computed the way the reference computes it -- ONE broadcast
    shift + mask over [K/P, P, N] and [G, N/P, P] in multithreaded torch CPU ops (qlinear_cuda_old.py:295-312,
    qlinear_cuda.py:257-277) instead of one numpy pass per field: what bench.py's cpu_baseline times, so that the port runs
    at the reference class's own speed.  3-bit layers (their unpack is a three-word splice, :313-344) take dequantize().
    Pinned bit-identical to dequantize() on every 2/4/8-bit fixture (tests/test_oracle_golden.py)."""
    if bits == 3:
        return dequantize(qweight, qzeros, scales, g_idx, bits, zero_mode)
    P, maxq = 32 // bits, (1 << bits) - 1
    wf = torch.arange(0, 32, bits, dtype=torch.int32)
    qw, qz, sc = qweight.cpu(), qzeros.cpu(), scales.cpu()
    w = torch.bitwise_and(torch.bitwise_right_shift(qw.unsqueeze(1).expand(-1, P, -1), wf.unsqueeze(-1)), maxq)   # [K/P, P, N] (arithmetic shift + mask = the field)
    K, N = w.shape[0] * P, w.shape[2]
    w = w.reshape(K, N)
    z = torch.bitwise_and(torch.bitwise_right_shift(qz.unsqueeze(2).expand(-1, -1, P), wf.unsqueeze(0)), maxq) + 1  # [G, N/P, P]
    if zero_mode == ZERO_WRAP:
        z = torch.bitwise_and(z, maxq)
    elif zero_mode != ZERO_NOWRAP:
        raise ValueError(f"unknown zero_mode {zero_mode!r}")
    z = z.reshape(z.shape[0], N)
    if g_idx is None:
        G = z.shape[0]
        gs = -(-K // G)
        if K == G * gs:                                            # sequential groups: broadcast instead of a gather (qlinear_cuda_old.py:346-349)
            diff = (w.reshape(G, gs, N) - z.unsqueeze(1)).to(sc.dtype)
            return (sc.unsqueeze(1) * diff).reshape(K, N)
        g = torch.from_numpy(default_g_idx(K, gs).astype(np.int64))
    else:
        g = torch.as_tensor(np.asarray(g_idx.cpu() if isinstance(g_idx, torch.Tensor) else g_idx), dtype=torch.long)
    return sc[g] * (w - z[g]).to(sc.dtype)


def forward_fast(x: torch.Tensor, qweight, qzeros, scales, g_idx, bias, bits: int, zero_mode: str) -> torch.Tensor:
    """forward() on dequantize_torch(): identical values, reference-speed unpack."""
    W = dequantize_torch(qweight, qzeros, scales, g_idx, bits, zero_mode)
    y = torch.matmul(x.reshape(-1, x.shape[-1]).cpu(), W).to(x.dtype).reshape(x.shape[:-1] + (W.shape[1],))
    if bias is not None:
        y = y + bias.cpu()
    return y


def forward(x: torch.Tensor, qweight, qzeros, scales, g_idx, bias, bits: int, zero_mode: str) -> torch.Tensor:
    """out = (x.reshape(-1,K) @ W).to(x.dtype).reshape(...) + bias  (qlinear_cuda_old.py:202-355)."""
    W = dequantize(qweight, qzeros, scales, g_idx, bits, zero_mode)
    out_shape = x.shape[:-1] + (W.shape[1],)
    y = torch.matmul(x.reshape(-1, x.shape[-1]).cpu(), W)
    y = y.to(x.dtype).reshape(out_shape)
    if bias is not None:
        y = y + bias.cpu()
    return y


def forward_f64(x, qweight, qzeros, scales, g_idx, bias, bits, zero_mode) -> torch.Tensor:
    """Same math with exact (un-rounded) dequant and fp64 accumulation: the 'true' answer that
    both the reference's fp16 path and the HIP kernels approximate."""
    w = unpack_weights(qweight, bits).astype(np.int32)
    z = unpack_zeros(qzeros, bits, zero_mode)
    K = w.shape[0]
    g = default_g_idx(K, -(-K // z.shape[0])) if g_idx is None else (
        g_idx.cpu().numpy() if isinstance(g_idx, torch.Tensor) else np.asarray(g_idx))
    g = g.astype(np.int64)
    W = scales.cpu().double()[torch.from_numpy(g)] * torch.from_numpy((w - z[g]).astype(np.float64))
    y = x.reshape(-1, x.shape[-1]).cpu().double() @ W
    if bias is not None:
        y = y + bias.cpu().double()
    return y.reshape(x.shape[:-1] + (W.shape[1],))


# --------------------------------------------------------------------------- #
# pack (float weights -> packed ints), bit-exact target
# --------------------------------------------------------------------------- #
def pack_rows(vals: np.ndarray, bits: int) -> np.ndarray:
    """Inverse of unpack_rows: uint32 [V, C] (fields OR-ed *unmasked*, exactly like the
    reference: an out-of-range value bleeds into its neighbours, qlinear_cuda.py:141,177 and
    SURVEY App. B #2) -> int32 [V/32*bits, C]."""
    assert bits in SUPPORTED_BITS
    v = vals.astype(np.uint32)
    vals_per_unit = 32 // bits if bits != 3 else 32
    words_per_unit = 1 if bits != 3 else 3
    assert v.shape[0] % vals_per_unit == 0
    units = v.reshape(v.shape[0] // vals_per_unit, vals_per_unit, *v.shape[1:])
    out = np.zeros((units.shape[0], words_per_unit) + v.shape[1:], dtype=np.uint32)
    if bits != 3:
        for j in range(vals_per_unit):
            out[:, 0] |= units[:, j] << np.uint32(bits * j)
    else:
        # reference order of operations, including its partial masks on the straddlers only
        for j in range(10):
            out[:, 0] |= units[:, j] << np.uint32(3 * j)
        out[:, 0] |= units[:, 10] << np.uint32(30)
        out[:, 1] |= (units[:, 10] >> np.uint32(2)) & np.uint32(1)
        for j in range(10):
            out[:, 1] |= units[:, 11 + j] << np.uint32(3 * j + 1)
        out[:, 1] |= units[:, 21] << np.uint32(31)
        out[:, 2] |= (units[:, 21] >> np.uint32(1)) & np.uint32(3)
        for j in range(10):
            out[:, 2] |= units[:, 22 + j] << np.uint32(3 * j + 2)
    return out.reshape((units.shape[0] * words_per_unit,) + v.shape[1:]).view(np.int32)


def quantize_to_int(W: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, g_idx, out_dtype) -> np.ndarray:
    """intweight[k,n] = round((W[n,k] + (zero*scale)[g,n]) / scale_cast[g,n])  -> uint32 [K,N].

    ``zero*scale`` uses the *input-precision* scales, the divide uses the scales already cast to
    the layer dtype (qlinear_cuda.py:119-120,127).  W is [N,K]; scales/zeros
    are [N,G] as handed over by GPTQ (transposed inside, :117-118).
    """
    s = scales.t().contiguous()
    z = zeros.t().contiguous()
    sz = z * s                                       # [G,N] input precision
    s_cast = s.clone().to(out_dtype)                 # [G,N]
    g = torch.as_tensor(np.asarray(g_idx), dtype=torch.long)
    num = W.t() + sz[g]                              # [K,N]
    q = torch.round(num / s_cast[g]).to(torch.int)   # torch type promotion as in the reference
    return q.numpy().astype(np.uint32)


def pack(W: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, g_idx, bits: int,
         out_dtype=torch.float16):
    """Full ``QuantLinear.pack`` restatement. Returns (qweight int32 [K/32*bits,N],
    qzeros int32 [G,N/32*bits], scales_out [G,N] out_dtype)."""
    K = W.shape[1]
    if g_idx is None:
        G = scales.shape[1]
        g_idx = default_g_idx(K, -(-K // G))
    intw = quantize_to_int(W, scales, zeros, g_idx, out_dtype)
    qweight = pack_rows(intw, bits)
    zm1 = (zeros.t().contiguous() - 1).numpy().astype(np.uint32)      # qlinear_cuda.py:169-170
    qzeros = np.ascontiguousarray(pack_rows(np.ascontiguousarray(zm1.T), bits).T)
    scales_out = scales.t().contiguous().clone().to(out_dtype)
    return torch.from_numpy(qweight.copy()), torch.from_numpy(qzeros.copy()), scales_out


# --------------------------------------------------------------------------- #
# act-order helper: the row re-sequencing the HIP path derives at post_init
# (semantics of exllama's make_sequential, autogptq_extension/exllama/cuda_func/q4_matrix.cu:105-169:
#  a stable counting sort of k by g_idx)
# --------------------------------------------------------------------------- #
def sequential_permutation(g_idx) -> np.ndarray:
    g = g_idx.cpu().numpy() if isinstance(g_idx, torch.Tensor) else np.asarray(g_idx)
    return np.argsort(g, kind="stable").astype(np.int32)


# --------------------------------------------------------------------------- #
# synthetic-input recipes shared by tests / bench / golden generator
# --------------------------------------------------------------------------- #
def golden_recipe_inputs(k: int, n: int, dtype=torch.float16, group_size: int = 128, bits: int = 4):
    """The deterministic recipe of the reference's known-answer tests
    (tests/test_q4.py:1086-1112 and 1781-1796): CPU RNG, seed 42."""
    torch.manual_seed(42)
    qweight = torch.randint(-100, 100, size=(k // 32 * bits, n), dtype=torch.int32)
    G = -(-k // group_size)
    scales = torch.zeros((G, n), dtype=dtype) + 0.002
    qzeros = torch.zeros((G, n // 32 * bits), dtype=torch.int32)
    x = torch.rand(1, 1, k, dtype=torch.float16).to(dtype)
    return qweight, qzeros, scales, x


def random_quant_layer(K: int, N: int, bits: int, group_size: int, *, act_order: bool = False,
                       dtype=torch.float16, seed: int = 0, bias: bool = False):
    """Throughput-style fixture: fully random packed words (every bit pattern is a legal
    qweight/qzeros), scales ~ 0.002*(1+0.1*rand).  SURVEY §8(d)."""
    gen = torch.Generator().manual_seed(seed)
    gs = K if group_size == -1 else group_size
    G = -(-K // gs)
    qweight = torch.randint(-2**31, 2**31 - 1, (K // 32 * bits, N), dtype=torch.int64, generator=gen).to(torch.int32)
    qzeros = torch.randint(-2**31, 2**31 - 1, (G, N // 32 * bits), dtype=torch.int64, generator=gen).to(torch.int32)
    scales = (0.002 * (1 + 0.1 * torch.rand(G, N, generator=gen))).to(dtype)
    g_idx = torch.from_numpy(default_g_idx(K, gs))
    if act_order:
        g_idx = g_idx[torch.randperm(K, generator=gen)].contiguous()
    b = (0.1 * torch.randn(N, generator=gen)).to(dtype) if bias else None
    return dict(qweight=qweight, qzeros=qzeros, scales=scales, g_idx=g_idx, bias=b,
                bits=bits, group_size=gs, K=K, N=N)


def minmax_quantize(W: torch.Tensor, bits: int, group_size: int, g_idx=None, sym: bool = False):
    """Per-(group, column) asymmetric min/max quantizer (the shape of what GPTQ's Quantizer
    emits, auto_gptq/quantization/quantizer.py:56-96): returns scale [N,G], zero [N,G] (float
    tensors holding integer zero-points clamped to [1, maxq] so zero-1 is storable)."""
    N, K = W.shape
    gs = K if group_size == -1 else group_size
    g = default_g_idx(K, gs) if g_idx is None else np.asarray(g_idx)
    G = int(g.max()) + 1
    maxq = 2 ** bits - 1
    scale = torch.empty(N, G, dtype=W.dtype)
    zero = torch.empty(N, G, dtype=W.dtype)
    gt = torch.from_numpy(g.astype(np.int64))
    for gi in range(G):
        cols = W[:, gt == gi]
        lo = torch.clamp(cols.min(dim=1).values, max=0)
        hi = torch.clamp(cols.max(dim=1).values, min=0)
        if sym:
            m = torch.maximum(lo.abs(), hi)
            lo, hi = -m, m
        s = (hi - lo) / maxq
        s = torch.where(s == 0, torch.ones_like(s), s)
        scale[:, gi] = s
        zero[:, gi] = torch.clamp(torch.round(-lo / s), 1, maxq) if not sym else torch.full_like(s, (maxq + 1) / 2)
    return scale, zero



# --------------------------------------------------------------------------- #
# The decode copy (gptq_prepack_decode, include/gptq_mi355x.h): restated so the tests can pin the device kernels to it.
# Role in the reference: the load-time re-layouts of its fast backends (exllamav2 shuffle_kernel exllamav2/cuda/q_matrix.cu:19-42, Marlin's repack
# marlin/marlin_repack.cu:8-92).  The layout is this library's own; what is pinned to the REFERENCE is that it is lossless: the reference's integer
# unpack of the inverse equals its unpack of the checkpoint tensor, and the constants are the reference's scales / zero-points.
# --------------------------------------------------------------------------- #
_PAIR_ORDER = (0, 2, 4, 6, 1, 3, 5, 7)      # stored nibble p holds source nibble _PAIR_ORDER[p]


_BYTE_ORDER = (0, 2, 1, 3)                  # 8-bit: stored byte p holds source value _BYTE_ORDER[p]


def _decode_copy_lane_words(v: np.ndarray, bits: int) -> np.ndarray:
    """v: uint32 [..., KPL] = the KPL consecutive k a lane owns (32; 16 at 8 bits) -> uint32 [..., WPL] stored words (include/gptq_mi355x.h, qweight_tiled)."""
    if bits == 2:                                            # two words: word w holds pair p = (k 16 w + 2 p, k 16 w + 2 p + 1) at bit 2 p of its low / high half
        out = np.zeros(v.shape[:-1] + (2,), dtype=np.uint32)
        for w in range(2):
            for p in range(8):
                out[..., w] |= v[..., 16 * w + 2 * p] << np.uint32(2 * p)
                out[..., w] |= v[..., 16 * w + 2 * p + 1] << np.uint32(16 + 2 * p)
        return out
    if bits == 4:
        out = np.zeros(v.shape[:-1] + (4,), dtype=np.uint32)
        for w in range(4):
            for p, src in enumerate(_PAIR_ORDER):
                out[..., w] |= v[..., 8 * w + src] << np.uint32(4 * p)
        return out
    if bits == 8:
        out = np.zeros(v.shape[:-1] + (4,), dtype=np.uint32)
        for w in range(4):
            for p, src in enumerate(_BYTE_ORDER):
                out[..., w] |= v[..., 4 * w + src] << np.uint32(8 * p)
        return out
    assert bits == 3
    out = np.zeros(v.shape[:-1] + (3,), dtype=np.uint32)
    for j in range(3):
        for i in range(5):
            pr = 5 * j + i                                   # pair pr = (k 2 pr, k 2 pr + 1)
            out[..., j] |= v[..., 2 * pr] << np.uint32(3 * i)
            out[..., j] |= v[..., 2 * pr + 1] << np.uint32(16 + 3 * i)
        out[..., j] |= ((v[..., 30] >> np.uint32(j)) & np.uint32(1)) << np.uint32(15)
        out[..., j] |= ((v[..., 31] >> np.uint32(j)) & np.uint32(1)) << np.uint32(31)
    return out


def decode_copy_weights(qweight: torch.Tensor, bits: int = 4) -> torch.Tensor:
    """int32 [K/32*bits, N] -> int32 [N/16, chunks, 4 (k-slot), 16 (column), WPL (word)]: the lane (kb, col) of chunk c of strip s holds the KPL
    consecutive k from c * 4 KPL + kb * KPL of column 16 s + col (KPL = 32, 16 at 8 bits; WPL = 4, 3 at 3 bits, 2 at 2 bits), re-encoded per
    _decode_copy_lane_words; k past K read as 0.  At 4 bits this is nibble_pair_order(qweight[16c + 4kb + w, 16s + col])."""
    assert bits in (2, 3, 4, 8)
    w = unpack_weights(qweight, bits).astype(np.uint32)                  # [K, N]
    K, N = w.shape
    kpl = 16 if bits == 8 else 32
    cke = 4 * kpl
    chunks = -(-K // cke)
    pad = np.zeros((chunks * cke, N), dtype=np.uint32)
    pad[:K] = w
    v = pad.reshape(chunks, 4, kpl, N // 16, 16).transpose(3, 0, 1, 4, 2)    # [s, c, kb, col, k in lane]
    return torch.from_numpy(np.ascontiguousarray(_decode_copy_lane_words(np.ascontiguousarray(v), bits)).view(np.int32))


def decode_copy_weights_inverse(tiled: torch.Tensor, K: int) -> torch.Tensor:
    """The checkpoint layout int32 [K/8, N] back from a decode copy."""
    t = tiled.cpu().numpy().view(np.uint32)
    S, chunks = t.shape[0], t.shape[1]
    sh = np.ascontiguousarray(t.transpose(1, 2, 4, 0, 3)).reshape(chunks * 16, S * 16)     # [c, kb, w, s, col] -> rows, columns
    q = np.zeros_like(sh)
    for p, src in enumerate(_PAIR_ORDER):
        q |= ((sh >> np.uint32(4 * p)) & np.uint32(15)) << np.uint32(4 * src)
    return torch.from_numpy(np.ascontiguousarray(q[:K // 8]).view(np.int32))


def decode_copy_consts(qzeros: torch.Tensor, scales: torch.Tensor, zero_mode: str, bits: int = 4) -> torch.Tensor:
    """uint8 [N/16, G, REC]: 16 scales (bit copies, 2 bytes each, little endian) at byte 0, then the 16 zero-points as used in dequant from byte 32:
    one byte each (REC = 48) at 2 / 3 / 4 bits, two bytes each (REC = 64) at 8 bits, where nowrap reaches 256."""
    z = unpack_zeros(qzeros, bits, zero_mode)                                               # [G, N]
    G, N = z.shape
    sb = scales.cpu().contiguous().view(torch.int16).numpy().view(np.uint8).reshape(G, N, 2)
    rec = 64 if bits == 8 else 48
    out = np.zeros((N // 16, G, rec), dtype=np.uint8)
    out[:, :, :32] = sb.reshape(G, N // 16, 32).transpose(1, 0, 2)
    if bits == 8:
        zb = z.astype(np.uint16).reshape(G, N, 1).view(np.uint8).reshape(G, N // 16, 32)
        out[:, :, 32:] = zb.transpose(1, 0, 2)
    else:
        out[:, :, 32:] = z.astype(np.uint8).reshape(G, N // 16, 16).transpose(1, 0, 2)
    return torch.from_numpy(out)
