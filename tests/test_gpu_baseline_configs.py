"""GPU (-m gpu): parity at the shapes BASELINE.json's configs name, not at scaled-down stand-ins.

config 3  Llama-7B shapes, int4 g128 desc_act=True, M = 2048 (all three layer shapes)
config 4  Llama-2-70B TP=8 shards: column shards 8192->1024, 8192->3584, 28672->1024; row shard 3584->8192
config 5  int3 / int8 group_size=32 on the Llama-7B shapes, decode (GEMV, M = 1/4/8), batched decode
          (M = 16/64) and prefill (tiled MFMA GEMM, M = 2048), fp16 and bf16, act-order on and off

Every case is checked four ways, all against the ORACLE (oracle/gptq_oracle.py: the reference's unpack + dequant),
never against another kernel of this library:
  (0) the library's full-size dequantize() equals the oracle's dequantised weight bit for bit;
  (a) EVERY one of the M x N outputs against x (fp64) @ W_oracle (fp64) -- the oracle's weight, multiplied in fp64 on
      the GPU by torch (plumbing: milliseconds), so every row tile x column tile of every launch geometry is compared;
  (b) one-hot rows of x -- at least one in every 32-row tile -- return the oracle's dequantised weight rows EXACTLY
      (bit for bit the reference's scales * (weight - zeros), qlinear_cuda_old.py:348 / qlinear_cuda.py:302) through
      whatever kernel the planner picks for that M;
  (c) bit reproducibility of a repeated call.
Tolerance (a): |y - ref| <= atol * max|ref| + rtol * |ref| with (rtol, atol) = (1e-3, 1e-3) fp16, (8e-3, 8e-3) bf16 -- the scale of
the reference's own atol = 2e-2 on O(7) outputs (tests/test_q4.py:1120,1802), with no sqrt(K) allowance.
Reference test this mirrors in spirit: tests/test_hpu_linear.py:102-181 (shape x dtype x pattern grid) and
tests/test_q4.py:1060-1122 (kernel output vs the Python path).
"""
import pytest
import torch

from autogptq_amd.qlinear_mi355x import QuantLinear
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = {torch.float16: (1e-3, 1e-3), torch.bfloat16: (8e-3, 8e-3)}          # (rtol, atol relative to the output scale)
CHECKED = {"cases": 0, "outputs": 0}                                       # run count, printed at session end (conftest)
LLAMA7B = [(4096, 4096), (4096, 11008), (11008, 4096)]

_LAYERS = {}


def _layer(bits, gs, K, N, act, dtype):
    key = (bits, gs, K, N, act, dtype)
    if key not in _LAYERS:
        _LAYERS.clear()                       # one full-size layer alive at a time
        L = O.random_quant_layer(K, N, bits, gs, act_order=act, dtype=dtype, seed=bits * 131 + K // 64 + N // 32 + int(act))
        q = QuantLinear(bits, gs, K, N, False, weight_dtype=dtype)
        q.qweight, q.qzeros, q.scales, q.g_idx = L["qweight"].clone(), L["qzeros"].clone(), L["scales"].clone(), L["g_idx"].clone()
        q = q.to(DEV)
        q.post_init()
        mode = O.reference_zero_mode(act, bits)
        W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], bits, mode)      # the ORACLE's weight, CPU, [K, N] dtype
        with torch.no_grad():
            Wlib = q.dequantize()
        assert torch.equal(Wlib.cpu(), W), f"full-size dequantize() differs from the oracle (bits={bits} {K}x{N} act={act} {dtype})"    # (0)
        del Wlib
        Wd = W.to(DEV)
        _LAYERS[key] = (L, q, Wd, Wd.double())
    return _LAYERS[key]


def _slice_cols(L, bits, n0, n1):
    z0, z1 = n0 * bits // 32, n1 * bits // 32
    return L["qweight"][:, n0:n1], L["qzeros"][:, z0:z1], L["scales"][:, n0:n1]


def _hot_rows(M, K, gs):
    """(row of x, k) pairs: one one-hot row in every 32-row tile (all of them for M <= 4), k spread over K with the
    edge cases first (first / last k, a group edge, a 3-bit straddler)."""
    if M == 1:
        return []
    rows = list(range(min(M, 4))) if M <= 32 else [t * 32 + (t * 7) % 32 for t in range(M // 32)]
    special = [0, K - 1, gs, 10]
    return [(r, special[i] if i < 4 else (i * 2654435761) % K) for i, r in enumerate(rows)]


def _check(bits, gs, K, N, M, act, dtype):
    L, q, W, W64 = _layer(bits, gs, K, N, act, dtype)
    mode = O.reference_zero_mode(act, bits)
    assert q.resolved_zero_mode() == int(mode == O.ZERO_NOWRAP)
    gen = torch.Generator().manual_seed(M * 7 + bits)
    x = (torch.rand(M, K, generator=gen) - 0.5).to(dtype)
    hot = _hot_rows(M, K, gs)
    for r, k in hot:
        x[r].zero_()
        x[r, k] = 1.0
    xd = x.to(DEV)
    with torch.no_grad():
        y, yb = q(xd), q(xd)
    assert y.shape == (M, N) and y.dtype == dtype
    assert torch.equal(y, yb), "not bit-reproducible"                       # (c)
    for r, k in hot:                                                        # (b)
        assert torch.equal(y[r], W[k]), f"one-hot row {r} -> k={k}: output is not the oracle's exact dequantised weight row (M={M})"
    # (a) every output against the oracle's weight in fp64
    ref = xd.double() @ W64
    rtol, atol = TOL[dtype]
    scale = max(1e-6, float(ref.abs().max()))
    diff = (y.double() - ref).abs()
    bad = diff > atol * scale + rtol * ref.abs()
    nbad = int(bad.sum())
    if nbad:
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"bits={bits} g={gs} {K}x{N} M={M} act={act} {dtype}: {nbad}/{bad.numel()} outputs out of tolerance, first at "
                             f"[{idx[0]}, {idx[1]}], max abs diff {float(diff.max())} (scale {scale})")
    CHECKED["cases"] += 1
    CHECKED["outputs"] += M * N


# ------------------------------------------------------------------------------------------ config 5
@pytest.mark.parametrize("act", [False, True], ids=["seq", "act"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("K,N", LLAMA7B)
@pytest.mark.parametrize("bits", [3, 8])
def test_config5_int3_int8_g32_llama7b_shapes(bits, K, N, dtype, act):
    """BASELINE config 5: int3 / int8, group_size 32, on the Llama-7B layer shapes -- GEMV-generic plans at full
    K (16 waves, U = 4 for int8, masked tail rows), strips / skinny at M = 16 / 64, and at M = 2048 the stream-K prefill kernel on the 3- / 8-bit decode
    copy (csrc/gemm_wide_sk_b38.hip: each half of the wave on its own 32-wide group; act-order: x permuted in natural order by the pre-pass)."""
    from autogptq_amd import _lib
    _, q, _, _ = _layer(bits, 32, K, N, act, dtype)
    plan = _lib.describe_plan(q._layer, 2048)
    assert plan["kernel"] == "wide_sk" and plan["perm"] == int(act) and plan["tiles"] == f"16x{N // 256}", plan
    for M in (1, 4, 8, 16, 64, 2048):
        _check(bits, 32, K, N, M, act, dtype)


# ------------------------------------------------------------------------------------------ config 3
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("K,N", LLAMA7B)
def test_config3_int4_g128_desc_act_prefill_2048(K, N, dtype):
    """BASELINE config 3: int4 g128 desc_act=True, seq_len 2048 prefill on all three Llama-7B shapes (the tiled MFMA
    kernel with the group-sorted weight copy and permuted x), plus the decode / batched-decode row counts."""
    from autogptq_amd import _lib
    _, q, _, _ = _layer(4, 128, K, N, True, dtype)
    plan = _lib.describe_plan(q._layer, 2048)
    # round 5: the stream-K form of the 128 x 128 wave tile on the decode copy (csrc/gemm_wide_sk.hip), x permuted in natural order by the pre-pass
    assert plan["kernel"] == "wide_sk" and plan["perm"] == 1 and plan["tiles"] == f"16x{N // 256}", plan
    for M in (2048, 1, 8, 64):
        _check(4, 128, K, N, M, True, dtype)


@pytest.mark.parametrize("act", [False, True], ids=["seq", "act"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("K,N", LLAMA7B)
def test_batched_rows_17_to_256_int4_g128_llama7b_shapes(K, N, dtype, act):
    """The row counts between decode and prefill (the reference's kernel switch sits at 8 / 50 rows) on the Llama-7B shapes with the DEFAULT plan: round 5's
    exchange-free kernel on the decode copy (csrc/gemm_rows.hip) from 5 to 64 rows, round 6's whole-K panel kernel (csrc/gemm_panel.hip) above."""
    from autogptq_amd import _lib
    _, q, _, _ = _layer(4, 128, K, N, act, dtype)
    for M in (5, 8, 16, 17, 33, 64, 96, 128, 192, 256):
        plan = _lib.describe_plan(q._layer, M)
        # round 6: from 96 rows (33 on the 11008-column layer: ONE row panel, partial below 64 rows, 172 tiles) the whole-K panel kernel (csrc/gemm_panel.hip) where its 64-row tiles fill the chip
        want = "panel" if ((M >= 96 and (K <= 8192 or M >= 160)) or (33 <= M <= 64 and N == 11008)) else "rows"      # (deep layers at few rows stay with the rows kernel)
        if act and K >= 8192 and M <= 16:
            want = "stream64"          # late round 6: act-order layers of K >= 8192 at up to 16 rows keep the 64-column-strip kernel (15.5 / 17.0 -> 14.2 / 15.3 us on this layer)
        assert plan["kernel"] == want, (M, plan)
        _check(4, 128, K, N, M, act, dtype)


@pytest.mark.parametrize("act", [False, True], ids=["seq", "act"])
@pytest.mark.parametrize("K,N,plans", [(13824, 5120, {24: "mid", 48: "rows", 96: "rows", 128: "panel", 192: "panel"}), (17920, 6656, {64: "rows", 96: "rows", 192: "panel"})])
def test_deep_down_projections_13b_30b_mid_band(K, N, plans, act):
    """Late round 6 (tools/mid_band_sweep.py): the DEEP layers of the larger families (K > 8192: Llama-13B / 30B down projections) at 33 ... 255 rows -- the exchange-free
    rows kernel beyond its 64 Mi-weight limit (33 .. 128 rows), the panel kernel beyond K = 16384 at 160 .. 255 rows; plan asserted, every output against the fp64
    oracle product, one-hot rows exact, bit-reproducible."""
    from autogptq_amd import _lib
    _, q, _, _ = _layer(4, 128, K, N, act, torch.float16)
    for M, want in plans.items():
        plan = _lib.describe_plan(q._layer, M)
        assert plan["kernel"] == want, (K, N, M, plan)
        _check(4, 128, K, N, M, act, torch.float16)


def test_north_star_m4096_4096x4096():
    """north_star's second target: batch x seq = 4096 rows on 4096 -> 4096, int4 g128."""
    _check(4, 128, 4096, 4096, 4096, False, torch.float16)


# ------------------------------------------------------------------------------------------ config 4
@pytest.mark.parametrize("K,N", [(8192, 1024), (8192, 3584), (28672, 1024)])
@pytest.mark.parametrize("act", [False, True], ids=["seq", "act"])
def test_config4_llama70b_tp8_column_shards(K, N, act):
    """BASELINE config 4: the per-GPU column shards of Llama-2-70B at TP = 8 (8192 -> 8192/8, 8192 -> 28672/8,
    28672 -> 8192/8), decode and prefill row counts."""
    for M in (1, 4, 64, 2048):
        _check(4, 128, K, N, M, act, torch.float16)


def test_config4_llama70b_row_shard_and_act_order_refusal():
    """Row-parallel pairing of the down projection: the K shard 28672/8 = 3584 -> 8192 is an ordinary layer; an
    act-order layer cannot be row-split (its g_idx mixes groups across the whole K) and the wrapper refuses."""
    from autogptq_amd.tensor_parallel import RowParallelQuantLinear

    for M in (1, 8, 2048):
        _check(4, 128, 3584, 8192, M, False, torch.float16)
    L = O.random_quant_layer(1024, 256, 4, 128, act_order=True, seed=3)
    full = QuantLinear(4, 128, 1024, 256, False)
    full.qweight, full.qzeros, full.scales, full.g_idx = L["qweight"], L["qzeros"], L["scales"], L["g_idx"]
    with pytest.raises(ValueError, match="sequential groups"):
        RowParallelQuantLinear.from_full(full, 0, 8)


# ------------------------------------------------------------------------------------------ config 2 (the headline)
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("K,N", LLAMA7B)
def test_config2_int4_g128_decode(K, N, dtype):
    """BASELINE config 2 -- the headline: int4 g128, NO act-order, decode row counts on the three Llama-7B shapes, every output
    of the planner's own kernel (register GEMV, streamed GEMV, 16-column strips / batched-decode kernels at 8 / 16 rows) against the
    fp64 oracle product.  Mirrors tests/test_q4.py:1060-1122 (kernel output vs the Python path on the layer's real shape)."""
    for M in (1, 2, 3, 4, 8, 16):
        _check(4, 128, K, N, M, False, dtype)


def _plain_layer(K, N, dtype, seed):
    L = O.random_quant_layer(K, N, 4, 128, act_order=False, dtype=dtype, seed=seed)
    q = QuantLinear(4, 128, K, N, False, weight_dtype=dtype)
    q.qweight, q.qzeros, q.scales, q.g_idx = L["qweight"].clone(), L["qzeros"].clone(), L["scales"].clone(), L["g_idx"].clone()
    q = q.to(DEV)
    q.post_init()
    W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 4, O.reference_zero_mode(False, 4))
    return q, W.to(DEV)


def _assert_all_outputs(y, xd, W, dtype, what):
    ref = xd.double() @ W.double()
    rtol, atol = TOL[dtype]
    scale = max(1e-6, float(ref.abs().max()))
    diff = (y.double() - ref).abs()
    bad = diff > atol * scale + rtol * ref.abs()
    nbad = int(bad.sum())
    if nbad:
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(f"{what}: {nbad}/{bad.numel()} outputs out of tolerance, first at [{idx[0]}, {idx[1]}], max abs diff "
                             f"{float(diff.max())} (scale {scale})")
    CHECKED["cases"] += 1
    CHECKED["outputs"] += y.numel()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("group", ["qkv", "gate_up"])
def test_headline_forward_multi_full_size(group, dtype):
    """The two multi-layer launches bench.py's headline times -- gptq_forward_multi on q|k|v (3 x 4096 -> 4096) and on gate|up
    (2 x 4096 -> 11008) -- at their real size: EVERY output of EVERY layer against x (fp64) @ W_oracle (fp64), a one-hot row returning the
    oracle's exact weight row, bit-reproducible, and the same again replayed from a captured hipGraph (what the bench times).
    The three / two layers are DIFFERENT random layers, so a strip that lands in the wrong layer or column cannot cancel out."""
    from autogptq_amd.qlinear_mi355x import forward_multi

    K, N, n = (4096, 4096, 3) if group == "qkv" else (4096, 11008, 2)
    layers, Ws = [], []
    for i in range(n):
        q, W = _plain_layer(K, N, dtype, seed=1000 + 17 * i + N // 64)
        layers.append(q)
        Ws.append(W)
    for M in (1, 2, 4):
        gen = torch.Generator().manual_seed(M * 11 + n)
        x = (torch.rand(M, K, generator=gen) - 0.5).to(dtype)
        hot_k = (K - 1, 129, 0, 2047)[:M]
        if M > 1:                                                           # row M-1 one-hot (M = 1 keeps its dense row)
            x[M - 1].zero_()
            x[M - 1, hot_k[M - 1]] = 1.0
        xd = x.to(DEV)
        with torch.no_grad():
            ys = forward_multi(layers, xd)
            ys2 = forward_multi(layers, xd)
        for i in range(n):
            assert ys[i].shape == (M, N) and ys[i].dtype == dtype
            assert torch.equal(ys[i], ys2[i]), f"{group}[{i}] M={M}: not bit-reproducible"
            if M > 1:
                assert torch.equal(ys[i][M - 1], Ws[i][hot_k[M - 1]]), f"{group}[{i}] M={M}: one-hot row is not the oracle's weight row"
            _assert_all_outputs(ys[i], xd, Ws[i], dtype, f"forward_multi {group}[{i}] {K}x{N} M={M} {dtype}")
        # the same launch replayed from a hipGraph, on fresh x (the graph holds the pointers: x is rewritten in place)
        xs = xd.clone()
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s), torch.no_grad():
            forward_multi(layers, xs)                                       # workspace of this stream allocated before capture
            s.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                yg = forward_multi(layers, xs)
        for rep in range(3):
            xs.copy_((torch.rand(M, K, generator=gen) - 0.5).to(dtype))
            g.replay()
            torch.cuda.synchronize()
            with torch.no_grad():
                ye = forward_multi(layers, xs)
            for i in range(n):
                assert torch.equal(yg[i], ye[i]), f"{group}[{i}] M={M}: graph replay {rep} differs from the eager call"
            if rep == 0:
                for i in range(n):
                    _assert_all_outputs(yg[i], xs, Ws[i], dtype, f"forward_multi graph replay {group}[{i}] M={M} {dtype}")
    del layers, Ws


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_headline_mlp_forward_full_size_with_intermediate(dtype):
    """The default gptq_mlp_forward (gate | up in one launch, SiLU * mul, down) on the Llama-7B MLP 4096 -> 11008 -> 4096: the INTERMEDIATE
    silu(g) * u (read back from the call's staging buffer at the front of the workspace body) is checked entry by entry against the fp64 oracle
    products of gate and up -- a handful of wrong gate / up columns cannot hide in the down projection's tolerance -- and the final
    output against down applied, in fp64, to the intermediate the kernel really produced."""
    from autogptq_amd import _lib
    from autogptq_amd import qlinear_mi355x as qm

    K, I = 4096, 11008
    gate, Wg = _plain_layer(K, I, dtype, seed=2001)
    up, Wu = _plain_layer(K, I, dtype, seed=2002)
    down, Wd = _plain_layer(I, K, dtype, seed=2003)
    esz = 2
    for M in (1, 4, 16):
        gen = torch.Generator().manual_seed(M + 5)
        xd = ((torch.rand(M, K, generator=gen) - 0.5) * 4).to(dtype).to(DEV)      # |gate| up to ~1: the SiLU is exercised off its linear part
        with torch.no_grad():
            y = qm.mlp_forward(gate, up, down, xd)
            y2 = qm.mlp_forward(gate, up, down, xd)
        torch.cuda.synchronize()
        assert torch.equal(y, y2), "mlp_forward: not bit-reproducible"
        ws = qm._WORKSPACE[(torch.cuda.current_device(), int(torch.cuda.current_stream().cuda_stream))][0]
        h = ws[_lib.WS_HEADER_BYTES:_lib.WS_HEADER_BYTES + M * I * esz].view(dtype).reshape(M, I).clone()
        g64 = xd.double() @ Wg.double()
        u64 = xd.double() @ Wu.double()
        href = torch.nn.functional.silu(g64) * u64
        rtol, atol = TOL[dtype]
        scale = max(1e-6, float(href.abs().max()))
        # silu(g) * u from fp32 sums rounded once to the layer dtype: the product of two results each within (rtol, atol) of its reference
        bad = (h.double() - href).abs() > 2 * atol * scale + 2 * rtol * href.abs()
        assert not bool(bad.any()), (f"mlp intermediate M={M} {dtype}: {int(bad.sum())}/{bad.numel()} entries out of tolerance, first at "
                                      f"{torch.nonzero(bad)[0].tolist()}")
        CHECKED["cases"] += 1
        CHECKED["outputs"] += h.numel()
        _assert_all_outputs(y, h, Wd, dtype, f"mlp_forward down M={M} {dtype}")
    del gate, up, down


# ------------------------------------------------------------------------------------------ north_star M = 4096: the 128 x 512 kernel
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("K,N,act", [(4096, 4096, True), (4096, 11008, False), (11008, 4096, False), (4096, 11008, True)])
def test_north_star_m4096_wide_tiles(K, N, act, dtype):
    """batch x seq = 4096 rows on the Llama-7B shapes: the launches where the planner takes gemm_wide_kernel (128 x 512 tiles, 128 x 128 per wave, accumulators
    in AGPRs) -- every output against the fp64 oracle product, a one-hot row in every 32-row tile returning the oracle's exact weight row, act-order through
    the slot-ordered permute pre-pass + DMA staging (fp16; bf16 act-order keeps the 128 x 256 kernel), a half-empty last column tile (11008 = 21.5 x 512)."""
    from autogptq_amd import _lib
    L, q, W, W64 = _layer(4, 128, K, N, act, dtype)
    plan = _lib.describe_plan(q._layer, 4096)
    # the layers carry their decode copy (act-order ones: of the re-sequenced rows): the wide kernel reads it and stages the raw -- for act-order layers the
    # naturally permuted -- x by LDS DMA
    assert plan["kernel"] == "wide_copy", plan
    _check(4, 128, K, N, 4096, act, dtype)


def test_wide_tiles_forced_on_ragged_shapes():
    """The same kernel forced (tuning.reserved[GPTQ_LAB_GEMM_VARIANT] = GPTQ_LAB_VARIANT_WIDE_ON) on shapes with a partial last row tile, a partial last column tile, bias, both zero-point conventions
    and group sizes 64 / 128 / 256 -- and identical, bit for bit, to the 128 x 256 kernel with one K group (same MFMA k order)."""
    from autogptq_amd import _lib
    for (K, N, gs, M, act) in ((256, 544, 128, 200, False), (512, 1056, 64, 333, True), (1024, 512, 256, 129, False), (384, 96, 128, 65, False)):
        for zm in ("auto", "nowrap"):
            Lq = O.random_quant_layer(K, N, 4, gs, act_order=act, seed=K + N + M, bias=True)
            q = QuantLinear(4, gs, K, N, True, zero_mode=zm)
            q.qweight, q.qzeros, q.scales, q.g_idx, q.bias = Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], Lq["bias"]
            q = q.to(DEV)
            q.post_init()
            mode = O.ZERO_NOWRAP if (zm == "nowrap" or act) else O.ZERO_WRAP
            W = O.dequantize(Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], 4, mode).to(DEV)
            x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).half().to(DEV)
            tw, tn, tc = _lib.GptqTuning(), _lib.GptqTuning(), _lib.GptqTuning()
            LAB = _lib.LAB
            tw.path, tw.reserved[LAB.GEMM_VARIANT], tw.ksplit = 3, LAB.VARIANT_WIDE_ROWS_ON, 1          # wide tiles on the checkpoint rows (register-staged x)
            tn.path, tn.reserved[LAB.GEMM_VARIANT], tn.ksplit = 3, LAB.VARIANT_ONE_K_GROUP, 1
            tc.path, tc.reserved[LAB.GEMM_VARIANT], tc.ksplit = 3, LAB.VARIANT_WIDE_ON, 1          # wide tiles, from the decode copy where the layer has one (raw x by LDS DMA)
            assert _lib.describe_plan(q._layer, M, tw)["kernel"] == "wide"
            has_copy = q._qweight_tiled is not None and K % 128 == 0
            assert _lib.describe_plan(q._layer, M, tc)["kernel"] == ("wide_copy" if has_copy else "wide")
            with torch.no_grad():
                yw, yw2, yn = q(x, tuning=tw), q(x, tuning=tw), q(x, tuning=tn)
                yc, yc2 = q(x, tuning=tc), q(x, tuning=tc)
            assert torch.equal(yw, yw2) and torch.equal(yc, yc2)
            assert torch.equal(yw, yn), f"128 x 512 and 128 x 256 tiles differ ({K}x{N} g{gs} M={M} act={act} {zm})"
            ref = x.double() @ W.double() + Lq["bias"].to(DEV).double()
            scale = float(ref.abs().max())
            for y, what in ((yw, "rows"), (yc, "copy")):           # the copy form sums k in another order (a half-wave takes 32 consecutive k): same tolerance, not the same bits
                bad = (y.double() - ref).abs() > 1e-3 * scale + 1e-3 * ref.abs()
                assert not bool(bad.any()), f"{K}x{N} g{gs} M={M} act={act} {zm} ({what}): {int(bad.sum())} outputs out of tolerance"
            hot = torch.zeros(M, K, dtype=torch.float16, device=DEV)
            rows = torch.arange(M, device=DEV)
            hot[rows, (rows * 37) % K] = 1.0                       # one-hot rows: the exact dequantised rows through the copy form too
            saved, q._layer.bias = q._layer.bias, None
            with torch.no_grad():
                yh = q(hot, tuning=tc)
            q._layer.bias = saved
            assert torch.equal(yh, W[(rows * 37) % K])
