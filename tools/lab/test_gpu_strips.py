"""GPU (-m gpu): batched decode (5 .. 64 rows) on the decode copy -- csrc/gemm_strips.hip: four 16-column strips per workgroup behind ONE staged x,
v_mfma_f32_16x16x32 on the copy's pair-ordered words, K quarters through LDS, K slices through granules.  Forced with tuning.reserved[GPTQ_LAB_GEMM_KERNEL] =
GPTQ_LAB_GEMM_STRIPS on shapes chosen for its seams and (where the planner prefers it) by default on the Llama-7B shapes (test_gpu_baseline_configs.py).

Every case: EVERY output against x (fp64) @ W_oracle (fp64) (+ bias), one-hot rows return the oracle's exact dequantised rows, repeated calls bit-identical.
Reference band: exllamav2 q_gemm.cu:118 (MAX_Q_GEMM_ROWS = 50), qlinear_cuda.py:34,212 (kernel_switch_threshold = 128); checked as tests/test_q4.py:1060-1122
checks its kernels."""
import pytest
import torch

from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import QuantLinear
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LAB = _lib.LAB


def _tune(ks=0):
    t = _lib.GptqTuning()
    t.path, t.ksplit = 3, ks
    t.reserved[LAB.GEMM_KERNEL] = LAB.GEMM_STRIPS
    return t


# (K, N, group_size, act_order, what the shape exercises)
CASES = [
    (1024, 192, 128, False, "3 column groups, 8 chunks: K slices by the planner"),
    (4096, 4096, 128, False, "Llama-7B attention shape: 64 column groups x 4 K slices"),
    (2048, 1056, 64, False, "66 strips: the last column group has two, groups of 64"),
    (512, 64, 32, False, "one column group, groups of 32 (one per k-slot)"),
    (1152, 320, 128, True, "act-order: re-sequenced copy, x permuted by the pre-pass; 9 chunks (ragged quarters)"),
    (11008, 256, 128, False, "86 chunks: uneven K slices and quarters"),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}x{c[1]}g{c[2]}{'act' if c[3] else ''}" for c in CASES])
def test_strips_forced_every_output(case, dtype):
    K, N, gs, act, _ = case
    for zm in ("auto", "nowrap"):
        Lq = O.random_quant_layer(K, N, 4, gs, act_order=act, seed=K + N, bias=True, dtype=dtype)
        q = QuantLinear(4, gs, K, N, True, weight_dtype=dtype, zero_mode=zm)
        q.qweight, q.qzeros, q.scales, q.g_idx, q.bias = Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], Lq["bias"]
        q = q.to(DEV)
        q.post_init()
        mode = O.ZERO_NOWRAP if (zm == "nowrap" or act) else O.ZERO_WRAP
        W = O.dequantize(Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], 4, mode).to(DEV)
        rtol = 1e-3 if dtype == torch.float16 else 8e-3
        for M, ks in ((5, 0), (8, 0), (16, 0), (17, 0), (33, 0), (48, 0), (64, 0), (16, 1), (12, 2), (40, 8)):
            t = _tune(ks)
            plan = _lib.describe_plan(q._layer, M, t)
            if plan["kernel"] != "strips64":                    # this (M, K slices) does not fit the LDS: refused, as the planner says
                assert plan["tiles"] == "0x0", plan
                with pytest.raises(_lib.GptqError):
                    q(torch.zeros(M, K, dtype=dtype, device=DEV), tuning=t)
                continue
            x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype)
            hot = [(r, (r * 131 + 7) % K) for r in range(1, M, 3)]
            for r, k in hot:
                x[r].zero_()
                x[r, k] = 1.0
            x = x.to(DEV)
            with torch.no_grad():
                y, y2 = q(x, tuning=t), q(x, tuning=t)
            assert tuple(y.shape) == (M, N) and torch.equal(y, y2), "not bit-reproducible"
            ref = x.double() @ W.double() + Lq["bias"].to(DEV).double()
            scale = float(ref.abs().max())
            bad = (y.double() - ref).abs() > rtol * scale + rtol * ref.abs()
            assert not bool(bad.any()), f"{K}x{N} g{gs} M={M} ks={ks} act={act} {zm} {dtype} {plan}: {int(bad.sum())}/{bad.numel()} out of tolerance, first {torch.nonzero(bad)[0].tolist()}"
            saved, q._layer.bias = q._layer.bias, None
            with torch.no_grad():
                yh = q(x, tuning=t)
            q._layer.bias = saved
            for r, k in hot:
                assert torch.equal(yh[r], W[k]), f"one-hot row {r} -> k={k} is not the exact dequantised weight row (M={M} ks={ks})"
        from autogptq_amd import qlinear_mi355x as qm
        assert not qm.exchange_error(DEV)
