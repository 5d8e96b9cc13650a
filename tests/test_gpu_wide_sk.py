"""GPU (-m gpu): the stream-K form of the wide prefill tile (csrc/gemm_wide_sk.hip: persistent workgroups over (tile, K-chunk) units, two K parts per
workgroup summed through LDS, cut tiles finished by the holder of their head piece) -- forced with tuning.reserved[3] = GPTQ_LAB_VARIANT_WIDE_SK_ON on
shapes chosen for its seams, and by the planner's own rule on the BASELINE config-3 shapes (tests/test_gpu_baseline_configs.py asserts the plan name there).

Every case: EVERY output against x (fp64) @ W_oracle (fp64) (+ bias), bit reproducibility of a repeated call, one-hot rows return the oracle's exact
dequantised weight rows.  Reference behaviour this kernel answers: Marlin's stripe partition + cross-block reduction
(autogptq_extension/marlin/marlin_cuda_kernel.cu:234-300, :580-660), checked the way the reference checks its kernels (tests/test_q4.py:1060-1122)."""
import pytest
import torch

from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import QuantLinear
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
WIDE_SK_ON, WIDE_SK_OFF = _lib.LAB.VARIANT_WIDE_SK_ON, _lib.LAB.VARIANT_WIDE_SK_OFF          # include/gptq_mi355x_lab.h


def _tune(v):
    t = _lib.GptqTuning()
    t.path, t.reserved[_lib.LAB.GEMM_VARIANT] = 3, v
    return t


# (K, N, group_size, M, act_order, what the shape exercises)
CASES = [
    (2048, 256, 128, 128, False, "one tile, 8 units on 8 workgroups: one finisher adds 7 published pieces"),
    (1024, 512, 128, 256, False, "4 tiles x 4 units on 16 workgroups: every tile cut in 4"),
    (256, 1024, 128, 128 * 130, False, "520 one-unit tiles: two tiles per workgroup, no cut"),
    (512, 1056, 128, 128 * 86 - 77, True, "430 tiles x 2 units on 256 workgroups: cut tiles, shifted last row tile, partial last column tile, act-order"),
    (1024, 544, 64, 333, False, "group_size 64 (constants per step), ragged M and N"),
    (768, 2560, 256, 128 * 9, True, "three units per tile (an odd split of the two K parts' chunks), groups of 256, act-order"),
    (4096, 512, 128, 640, False, "deep K: 16 units per tile, 10 tiles on 128 workgroups"),
]


# the 3-bit decode copy and 32-wide groups (gemm_wide_sk_b38.hip; BASELINE config 5 is int3 / int8 at group_size 32): (bits, K, N, group_size, M, act_order)
CASES_B38 = [
    (3, 1024, 512, 128, 256, False, "3 bits, groups of 128: every tile cut in 4"),
    (3, 512, 1056, 32, 128 * 86 - 77, True, "3 bits, 32-wide groups (each half of the wave on its own group), act-order, shifted last row tile, partial last column tile"),
    (3, 768, 544, 64, 333, False, "3 bits, groups of 64, three units per tile, ragged M and N"),
    (3, 4096, 512, 32, 640, False, "3 bits g32, deep K"),
    (8, 1024, 512, 128, 256, False, "8 bits (three rotating 4-word register sets), groups of 128: every tile cut in 4"),
    (8, 512, 1056, 32, 128 * 86 - 77, True, "8 bits, 32-wide groups, act-order, shifted last row tile, partial last column tile"),
    (8, 768, 544, 64, 333, False, "8 bits, groups of 64, three units per tile (an odd number of two-step bodies between rotations), ragged M and N"),
    (8, 4096, 512, 32, 640, False, "8 bits g32, deep K"),
    (4, 1024, 544, 32, 333, False, "4 bits on 32-wide groups: constants from the checkpoint rows, one group per half"),
    (4, 2048, 256, 32, 128, True, "4 bits g32 act-order: one tile, one finisher adds 7 published pieces"),
]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("case", CASES_B38, ids=[f"int{c[0]}_{c[1]}x{c[2]}g{c[3]}M{c[4]}{'act' if c[5] else ''}" for c in CASES_B38])
def test_wide_sk_3bit_and_g32_every_output(case, dtype):
    _every_output(case[0], case[1:], dtype)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("case", CASES, ids=[f"{c[0]}x{c[1]}g{c[2]}M{c[3]}{'act' if c[4] else ''}" for c in CASES])
def test_wide_sk_forced_every_output(case, dtype):
    _every_output(4, case, dtype)


def _every_output(bits, case, dtype):
    K, N, gs, M, act, _ = case
    for zm in ("auto", "nowrap"):
        Lq = O.random_quant_layer(K, N, bits, gs, act_order=act, seed=K + N + M, bias=True, dtype=dtype)
        q = QuantLinear(bits, gs, K, N, True, weight_dtype=dtype, zero_mode=zm)
        q.qweight, q.qzeros, q.scales, q.g_idx, q.bias = Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], Lq["bias"]
        q = q.to(DEV)
        q.post_init()
        assert q._qweight_tiled is not None
        mode = O.ZERO_NOWRAP if (zm == "nowrap" or act or bits == 3) else O.ZERO_WRAP          # auto: qlinear_cuda_old's 3-bit branch and the act-order class do not wrap
        W = O.dequantize(Lq["qweight"], Lq["qzeros"], Lq["scales"], Lq["g_idx"], bits, mode).to(DEV)
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(M)) - 0.5).to(dtype).to(DEV)
        t = _tune(WIDE_SK_ON)
        plan = _lib.describe_plan(q._layer, M, t)
        assert plan["kernel"] == "wide_sk", plan
        with torch.no_grad():
            y, y2 = q(x, tuning=t), q(x, tuning=t)
        assert torch.equal(y, y2), "not bit-reproducible"
        ref = x.double() @ W.double() + Lq["bias"].to(DEV).double()
        rtol = 1e-3 if dtype == torch.float16 else 8e-3
        scale = float(ref.abs().max())
        bad = (y.double() - ref).abs() > rtol * scale + rtol * ref.abs()
        assert not bool(bad.any()), f"int{bits} {K}x{N} g{gs} M={M} act={act} {zm} {dtype}: {int(bad.sum())}/{bad.numel()} outputs out of tolerance, first {torch.nonzero(bad)[0].tolist()}"
        hot = torch.zeros(M, K, dtype=dtype, device=DEV)
        rows = torch.arange(M, device=DEV)
        hot[rows, (rows * 37 + 5) % K] = 1.0                   # one-hot rows through every tile, piece and K part
        saved, q._layer.bias = q._layer.bias, None
        with torch.no_grad():
            yh = q(hot, tuning=t)
        q._layer.bias = saved
        assert torch.equal(yh, W[(rows * 37 + 5) % K]), "one-hot rows are not the exact dequantised weight rows"
        # the header words the kernel uses are zero again (flags of published pieces), the sticky error word untouched
        from autogptq_amd import qlinear_mi355x as qm
        torch.cuda.synchronize()
        for (buf, _) in qm._WORKSPACE.values():
            assert int(buf[:1024].view(torch.int32).abs().sum()) == 0, "a 'published' flag survived the launch"
            assert not qm.exchange_error(DEV)
