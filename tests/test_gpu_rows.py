"""GPU (-m gpu): the exchange-free batched-decode kernel (csrc/gemm_rows.hip: a workgroup = 16 RB rows x S strips of the decode copy x the whole K; the waves
split the 128-deep chunks and meet once through LDS) -- forced with tuning.reserved[3] = GPTQ_LAB_VARIANT_ROWS_ON and every (RB, S) geometry it is built for,
on shapes chosen for its seams: ragged row tiles (rows past M repeat the last row and are not stored), a last strip group that runs past N, fewer chunks than
waves, 32- / 64- / 128- / 256-wide groups and one group for the whole K, act-order (x permuted in natural order by the pre-pass), bias.

Every case: EVERY output against x (fp64) @ W_oracle (fp64) (+ bias), bit reproducibility of a repeated call, one-hot rows return the oracle's exact
dequantised weight rows.  Reference behaviour: the fused kernels the reference uses below its switch thresholds (qlinear_cuda.py:34,212;
exllamav2/cuda/q_gemm.cu:118), checked the way the reference checks its kernels (tests/test_q4.py:1060-1122)."""
import pytest
import torch

from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import QuantLinear
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _tune(rb, s):
    t = _lib.GptqTuning()
    t.path, t.reserved[_lib.LAB.GEMM_VARIANT] = 3, _lib.LAB.VARIANT_ROWS_ON
    t.reserved[0], t.reserved[1] = rb, s          # row blocks of 16, strips per workgroup
    return t


# (K, N, group_size, M, act_order, what the shape exercises)
CASES = [
    (512, 64, 128, 5, False, "4 chunks on 16 / 8 waves (most waves idle), 4 strips, 5 rows of one 16-row block"),
    (1024, 160, 128, 17, False, "10 strips: the last strip group runs past N for S = 3, 4, 6; 17 rows = two row tiles for RB = 1"),
    (2048, 256, 32, 33, False, "32-wide groups (one per k-slot), 33 rows"),
    (1536, 192, 64, 64, True, "64-wide groups, act-order, 12 chunks"),
    (4096, 96, 256, 100, False, "groups of 256 (two chunks per group), 100 rows, 6 strips"),
    (1280, 128, 1280, 48, True, "one group for the whole K, act-order"),
    (11008, 64, 128, 31, False, "86 chunks: uneven chunk ranges per wave"),
]
GEOMS = [(1, 1), (1, 2), (1, 3), (1, 4), (2, 1), (2, 2), (2, 3), (2, 4), (2, 6), (4, 1), (4, 2), (4, 3), (4, 4)]      # (4, S): the 64-row form (half chunks), 4 bits


def _geoms(bits):
    """(RB, S) the library builds per width: 6 strips only at 4 bits; 8 bits: 4 strips only with two row blocks (registers)."""
    return [(rb, s) for rb, s in GEOMS if not (s == 6 and bits != 4) and not (bits == 8 and rb == 1 and s == 4) and not (rb == 4 and bits != 4)]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("bits", [4, 3, 8])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}x{c[1]}g{c[2]}M{c[3]}{'act' if c[4] else ''}" for c in CASES])
def test_rows_forced_every_output(case, bits, dtype):
    K, N, gs, M, act, _ = case
    Lq = O.random_quant_layer(K, N, bits, gs, act_order=act, seed=K + N + M, bias=True, dtype=dtype)
    q = QuantLinear(bits, gs, K, N, True, weight_dtype=dtype)
    q.qweight, q.qzeros, q.scales, q.g_idx, q.bias = Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], Lq["bias"]
    q = q.to(DEV)
    q.post_init()
    assert q._qweight_tiled is not None
    mode = O.ZERO_NOWRAP if q.resolved_zero_mode() == 1 else O.ZERO_WRAP          # (one group for the whole K: a permuted g_idx of zeros is no act-order)
    W = O.dequantize(Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], bits, mode).to(DEV)
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype).to(DEV)
    ref = x.double() @ W.double() + Lq["bias"].to(DEV).double()
    rows = torch.arange(M, device=DEV)
    hot = torch.zeros(M, K, dtype=dtype, device=DEV)
    hot[rows, (rows * 37 + 5) % K] = 1.0
    rtol = 1e-3 if dtype == torch.float16 else 8e-3
    scale = float(ref.abs().max())
    for rb, s in _geoms(bits):
        t = _tune(rb, s)
        plan = _lib.describe_plan(q._layer, M, t)
        assert plan["kernel"] == "rows" and plan["mt"] == rb and plan["tiles"] == f"{-(-M // (16 * rb))}x{-(-(N // 16) // s)}", plan
        with torch.no_grad():
            y, y2 = q(x, tuning=t), q(x, tuning=t)
        assert torch.equal(y, y2), f"not bit-reproducible (RB={rb} S={s})"
        bad = (y.double() - ref).abs() > rtol * scale + rtol * ref.abs()
        assert not bool(bad.any()), f"int{bits} {K}x{N} g{gs} M={M} act={act} {dtype} RB={rb} S={s}: {int(bad.sum())}/{bad.numel()} outputs out of tolerance, first {torch.nonzero(bad)[0].tolist()}"
        saved, q._layer.bias = q._layer.bias, None
        with torch.no_grad():
            yh = q(hot, tuning=t)
        q._layer.bias = saved
        assert torch.equal(yh, W[(rows * 37 + 5) % K]), f"one-hot rows are not the exact dequantised weight rows (RB={rb} S={s})"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("bits,gs", [(4, 128), (3, 32), (8, 64)])
@pytest.mark.parametrize("act", [False, True], ids=["seq", "act"])
def test_rows_several_layers_in_one_launch(bits, gs, act, dtype):
    """gptq_forward_multi at 5 .. 64 rows: the strip groups of q|k|v (and of a [gate, up] pair of unequal widths) in ONE launch of the exchange-free kernel;
    act-order layers of one activation order read one permuted x.  Every output of every layer against the oracle product, repeat calls bit-identical, and the
    workspace query says what the path needs: nothing, or the header + the permuted x."""
    import ctypes
    from autogptq_amd.qlinear_mi355x import forward_multi
    t = _lib.GptqTuning()                                     # forced (lab knob 50): the planner's own rule takes only launches of 1024+ strips at 24 .. 64 rows
    t.path, t.reserved[_lib.LAB.GEMM_VARIANT] = 3, _lib.LAB.VARIANT_ROWS_ON
    K = 1024
    for widths in ((1024, 1024, 1024), (1536, 1024)):
        qs, Ws, bs = [], [], []
        g0 = None
        for i, N in enumerate(widths):
            Lq = O.random_quant_layer(K, N, bits, gs, act_order=act, seed=7 * N + i + bits, bias=True, dtype=dtype)
            if act:
                g0 = Lq["g_idx"] if g0 is None else g0
                Lq["g_idx"] = g0.clone()                      # one activation order for the group, as GPTQ produces q / k / v
            q = QuantLinear(bits, gs, K, N, True, weight_dtype=dtype)
            q.qweight, q.qzeros, q.scales, q.g_idx, q.bias = Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], Lq["bias"]
            q = q.to(DEV)
            q.post_init()
            mode = O.ZERO_NOWRAP if q.resolved_zero_mode() == 1 else O.ZERO_WRAP
            qs.append(q)
            Ws.append(O.dequantize(Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], bits, mode).to(DEV))
            bs.append(Lq["bias"].to(DEV))
        for M in (5, 16, 33, 64):
            x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype).to(DEV)
            with torch.no_grad():
                ys, ys2 = forward_multi(qs, x, t), forward_multi(qs, x, t)
            arr = (ctypes.POINTER(_lib.GptqLayer) * len(qs))(*[ctypes.pointer(q._layer) for q in qs])
            need = int(_lib.load().gptq_workspace_bytes_multi_ex(arr, len(qs), M, ctypes.byref(t)))
            assert need == (65536 + (M * K * 2 + 255) // 256 * 256 if act else 0), need          # = the exchange-free kernel took the launch
            rtol = 1e-3 if dtype == torch.float16 else 8e-3
            for y, y2, W, b in zip(ys, ys2, Ws, bs):
                assert torch.equal(y, y2), "not bit-reproducible"
                ref = x.double() @ W.double() + b.double()
                scale = float(ref.abs().max())
                bad = (y.double() - ref).abs() > rtol * scale + rtol * ref.abs()
                assert not bool(bad.any()), f"int{bits} g{gs} widths={widths} M={M} act={act} {dtype}: {int(bad.sum())}/{bad.numel()} outputs out of tolerance"
