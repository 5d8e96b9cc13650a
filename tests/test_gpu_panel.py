"""GPU (-m gpu): the whole-K panel kernel (csrc/gemm_panel.hip: 64 x 32 NT workgroup tiles over the whole K, the waves of a workgroup are K
parts that meet once through LDS, nothing exchanged between workgroups) -- forced with tuning.reserved[3] = GPTQ_LAB_VARIANT_PANEL_ON in every tile geometry
on shapes chosen for its seams, and by the planner's own rule at the row counts of the band (129 ... 767) on three shapes.

Every case: EVERY output against x (fp64) @ W_oracle (fp64) (+ bias), bit reproducibility of a repeated call, one-hot rows return the oracle's exact
dequantised weight rows.  Reference behaviour this band answers: exllama / exllamav2 switch to dequant + cuBLAS above their row thresholds
(autogptq_extension/exllamav2/cuda/q_gemm.cu:104-181, exllama/cuda_func/q4_matmul.cu:225-260), Marlin runs its stripe partition at any M
(marlin/marlin_cuda_kernel.cu:234-300); checked the way the reference checks its kernels (tests/test_q4.py:1060-1122)."""
import pytest
import torch

from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import QuantLinear
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PANEL_ON = _lib.LAB.VARIANT_PANEL_ON          # include/gptq_mi355x_lab.h


def _tune(geom=0):
    t = _lib.GptqTuning()
    t.path, t.reserved[_lib.LAB.GEMM_VARIANT], t.reserved[0] = 3, PANEL_ON, geom
    return t


# (K, N, group_size, M, act_order, what the shape exercises)
CASES = [
    (1024, 256, 128, 64, False, "one row tile, 16 steps on 8 waves"),
    (512, 544, 128, 129, False, "shifted last row tile (one own row), partial last column tile, one step per wave"),
    (256, 1024, 64, 333, True, "fewer steps (4) than waves (8): empty K parts; groups of 64 (constants every step); act-order; ragged M"),
    (2048, 96, 256, 200, False, "groups of 256 (one group over four steps), N = 3 column blocks"),
    (768, 160, 768, 767, True, "one group over the whole K, 12 steps on 8 waves (uneven K parts), N = 5 column blocks, act-order"),
    (4096, 128, 128, 256, False, "deep K: 64 steps"),
    (128, 32, 128, 64, False, "the smallest legal layer: two steps (six idle waves), ONE column block (wider tiles re-read block 0 for their other blocks and store nothing for them)"),
    (128, 4128, 64, 1000, True, "two steps, 129 column blocks, 16 row tiles, act-order"),
    (16384, 64, 128, 100, False, "very deep K: 256 steps, 32 per wave"),
    (512, 1024, 128, 1500, False, "24 row tiles (the planner takes 8192x1024 up to 1536 rows), shifted last tile"),
]
GEOMS = [21, 22, 23, 24]          # 20 + NT: 64 x 32 / 64 / 96 / 128 tiles


def _every_output(q, Lq, W, M, K, dtype, t, what):
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype).to(DEV)
    with torch.no_grad():
        y, y2 = q(x, tuning=t), q(x, tuning=t)
    assert torch.equal(y, y2), f"{what}: not bit-reproducible"
    ref = x.double() @ W.double() + Lq["bias"].to(DEV).double()
    rtol = 1e-3 if dtype == torch.float16 else 8e-3
    scale = float(ref.abs().max())
    bad = (y.double() - ref).abs() > rtol * scale + rtol * ref.abs()
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} outputs out of tolerance, first {torch.nonzero(bad)[0].tolist()}"
    hot = torch.zeros(M, K, dtype=dtype, device=DEV)
    rows = torch.arange(M, device=DEV)
    hot[rows, (rows * 37 + 5) % K] = 1.0                   # one-hot rows through every tile and K part
    saved, q._layer.bias = q._layer.bias, None
    with torch.no_grad():
        yh = q(hot, tuning=t)
    q._layer.bias = saved
    assert torch.equal(yh, W[(rows * 37 + 5) % K]), f"{what}: one-hot rows are not the exact dequantised weight rows"


def _layer(K, N, gs, act, dtype, zm, seed, bits=4):
    Lq = O.random_quant_layer(K, N, bits, gs, act_order=act, seed=seed, bias=True, dtype=dtype)
    q = QuantLinear(bits, gs, K, N, True, weight_dtype=dtype, zero_mode=zm)
    q.qweight, q.qzeros, q.scales, q.g_idx, q.bias = Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], Lq["bias"]
    q = q.to(DEV)
    q.post_init()
    assert q._qweight_tiled is not None
    # 'auto' = the convention of the reference class the module stands in for (cuda_old wraps, the act-order class does not); a g_idx of ONE group is the
    # default order whatever was asked for, so the module's own answer is taken, not the case's flag
    mode = O.ZERO_NOWRAP if q.resolved_zero_mode() == _lib.ZERO_NOWRAP else O.ZERO_WRAP
    W = O.dequantize(Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], bits, mode).to(DEV)
    return q, Lq, W


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}x{c[1]}g{c[2]}M{c[3]}{'act' if c[4] else ''}" for c in CASES])
def test_panel_forced_every_geometry_every_output(case, dtype):
    K, N, gs, M, act, _ = case
    for zm in ("auto", "nowrap"):
        q, Lq, W = _layer(K, N, gs, act, dtype, zm, K + N + M)
        for geom in GEOMS:
            t = _tune(geom)
            plan = _lib.describe_plan(q._layer, M, t)
            assert plan["kernel"] == "panel", plan
            assert plan["tiles"] == f"{-(-M // 64)}x{-(-N // (32 * (geom % 10)))}", plan
            _every_output(q, Lq, W, M, K, dtype, t, f"{K}x{N} g{gs} M={M} act={act} {zm} {dtype} geom {geom}")


# the other packings of the decode copy and 32-wide groups (BASELINE config 5 is int3 / int8 at group_size 32): (bits, K, N, group_size, M, act_order)
CASES_B38 = [
    (3, 512, 544, 32, 129, False, "3 bits, 32-wide groups (each half of the wave on its own group, constants every step), shifted last row tile, partial last column tile"),
    (3, 1024, 256, 128, 200, True, "3 bits, groups of 128 (constants every second step), act-order"),
    (3, 768, 160, 64, 333, False, "3 bits, groups of 64, 12 steps on 8 waves"),
    (8, 512, 544, 32, 129, False, "8 bits (two 16-byte loads per column block and step), 32-wide groups"),
    (8, 1024, 256, 128, 200, True, "8 bits, groups of 128, act-order"),
    (8, 768, 160, 64, 333, False, "8 bits, groups of 64"),
    (4, 1024, 544, 32, 333, True, "4 bits on 32-wide groups, act-order"),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("case", CASES_B38, ids=[f"int{c[0]}_{c[1]}x{c[2]}g{c[3]}M{c[4]}{'act' if c[5] else ''}" for c in CASES_B38])
def test_panel_3bit_8bit_and_g32_every_geometry_every_output(case, dtype):
    bits, K, N, gs, M, act, _ = case
    for zm in ("auto", "nowrap"):
        q, Lq, W = _layer(K, N, gs, act, dtype, zm, K + N + M + bits, bits=bits)
        for geom in (GEOMS[:3] if bits == 8 else GEOMS):          # (8 bits: up to three column blocks per tile)
            t = _tune(geom)
            plan = _lib.describe_plan(q._layer, M, t)
            assert plan["kernel"] == "panel" and plan["tiles"] == f"{-(-M // 64)}x{-(-N // (32 * (geom % 10)))}", plan
            _every_output(q, Lq, W, M, K, dtype, t, f"int{bits} {K}x{N} g{gs} M={M} act={act} {zm} {dtype} geom {geom}")


# the band itself, by the planner's rule (plan asserted): the review's row counts on three shapes, plain and act-order
BAND_SHAPES = [(4096, 4096), (4096, 11008), (2048, 5120)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("act", [False, True], ids=["plain", "act"])
@pytest.mark.parametrize("shape", BAND_SHAPES, ids=[f"{k}x{n}" for k, n in BAND_SHAPES])
def test_panel_band_default_plan_every_output(shape, act, dtype):
    K, N = shape
    q, Lq, W = _layer(K, N, 128, act, dtype, "auto", K + N)
    for M in (129, 192, 256, 384, 512, 767):
        plan = _lib.describe_plan(q._layer, M, None)
        # the plan the rule gives (panel_pays, CPU-tested in test_host_logic.py): the panel kernel over the whole band, except where 512+ rows on the 45 M-weight layer are
        # several rounds of its tiles and the stream-K kernel keeps them
        want = "wide_sk" if (N == 11008 and M >= 512) else "panel"
        assert plan["kernel"] == want, (K, N, M, plan)
        _every_output(q, Lq, W, M, K, dtype, None, f"{K}x{N} M={M} act={act} {dtype} default plan {plan['kernel']}")


# ONE partial panel (33 .. 63 rows): the x DMAs past the last row re-read it (rows 8 min(i, imax) + min(r8, rlast)), the rows past M are computed on a copy of
# row M - 1 and never stored.  x is allocated EXACTLY M rows inside a guard: the rows behind it hold NaN, and a read past row M - 1 would show as NaN in a stored
# output only if it were a row < M -- so the check on reads is the address arithmetic itself (every output, one-hot rows) plus rows_here = 33 .. 63 over every residue
# of 8.  (bits, K, N, group_size, act_order)
PARTIAL = [(4, 1024, 256, 128, False), (4, 512, 544, 64, True), (3, 1024, 256, 32, False), (8, 768, 160, 64, True), (4, 4096, 128, 128, False)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("case", PARTIAL, ids=[f"int{c[0]}_{c[1]}x{c[2]}g{c[3]}{'act' if c[4] else ''}" for c in PARTIAL])
def test_panel_partial_single_panel_every_output(case, dtype):
    bits, K, N, gs, act = case
    q, Lq, W = _layer(K, N, gs, act, dtype, "auto", K + N + bits + 1, bits=bits)
    for M in (17, 24, 25, 32, 33, 34, 39, 40, 41, 47, 48, 55, 56, 57, 63):
        for geom in (21, 22, 23):                                  # (four column blocks: 64+ rows only)
            t = _tune(geom)
            plan = _lib.describe_plan(q._layer, M, t)
            assert plan["kernel"] == "panel" and plan["tiles"] == f"1x{-(-N // (32 * (geom % 10)))}", plan
            _every_output(q, Lq, W, M, K, dtype, t, f"int{bits} {K}x{N} g{gs} M={M} act={act} {dtype} geom {geom} (partial panel)")
    assert _lib.describe_plan(q._layer, 16, _tune(21))["kernel"] != "panel"           # 16 rows: not this kernel's (one 16-row tile of the rows kernel)
    if bits != 8:
        assert _lib.describe_plan(q._layer, 48, _tune(24))["kernel"] != "panel"       # four column blocks: 64+ rows


@pytest.mark.parametrize("bits,gs", [(4, 128), (3, 32), (8, 32)])
def test_panel_partial_panel_is_the_default_on_the_wide_layer(bits, gs):
    """33 .. 63 rows on 4096 -> 11008 (172 tiles of 64 x 64 in ONE round where the rows kernel needs two rounds of row tiles): planned, every output; 32 rows and
    the square layer keep the rows kernel."""
    q, Lq, W = _layer(4096, 11008, gs, False, torch.float16, "auto", 11, bits=bits)
    for M in (33, 48, 63):
        plan = _lib.describe_plan(q._layer, M, None)
        assert plan["kernel"] == "panel" and plan["tiles"].startswith("1x"), (M, plan)
        _every_output(q, Lq, W, M, 4096, torch.float16, None, f"int{bits} g{gs} 4096x11008 M={M} default plan")
    assert _lib.describe_plan(q._layer, 32, None)["kernel"] == "rows"
    q2, _, _ = _layer(4096, 4096, gs, False, torch.float16, "auto", 12, bits=bits)
    assert _lib.describe_plan(q2._layer, 48, None)["kernel"] == "rows"
