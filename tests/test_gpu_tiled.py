"""GPU (-m gpu): the load-time decode copy of a 3-, 4- or 8-bit layer (gptq_prepack_decode: qweight_tiled + qconst_tiled) and the decode kernel that streams it
(gemv_tiled_kernel, csrc/gemv_tiled.hip).

What is pinned, all against the ORACLE (oracle/gptq_oracle.py), never against another kernel of this library:
  * gptq_prepack_decode writes exactly what oracle.decode_copy_weights / decode_copy_consts state (bit for bit, ragged K, every group size, both zero-point
    conventions, fp16 / bf16 scales), and the oracle's integer unpack of the inverse equals its unpack of the checkpoint tensor;
  * post_init leaves the checkpoint tensors bit-identical and out of the side buffer's way (state_dict keys unchanged);
  * every output of the tiled kernel against x (fp64) @ W_oracle (fp64) -- default plan and forced (waves, chunks per wave)
    geometries, 1..4 rows of x, fp16 / bf16, both zero-point conventions, bias, group sizes 32 / 64 / 128 / 256, K with a ragged last chunk,
    in-launch K slices, 2..4 layers of different widths in one launch; one-hot rows return the oracle's exact dequantised rows; repeated
    calls are bit-identical.
The reference's counterpart: tests/test_q4.py:1060-1122 (kernel output against the Python path), test_repacking.py:53-104 (a repacked layout
must give the same layer)."""
import ctypes

import pytest
import torch

from autogptq_amd import _lib
from autogptq_amd.qlinear_mi355x import QuantLinear, forward_multi
from oracle import gptq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = {torch.float16: (1e-3, 1e-3), torch.bfloat16: (8e-3, 8e-3)}


def _tune(waves=0, u=0, ks=0):
    t = _lib.GptqTuning()
    t.path = 8
    t.waves, t.ksplit = waves, ks
    t.reserved[_lib.LAB.DEPTH] = u
    return t


def _layer(K, N, gs, dtype, seed, zero_mode="auto", bias=False, bits=4, act=False):
    L = O.random_quant_layer(K, N, bits, gs, dtype=dtype, seed=seed, bias=bias, act_order=act)
    q = QuantLinear(bits, gs, K, N, bias, weight_dtype=dtype, zero_mode=zero_mode)
    q.qweight, q.qzeros, q.scales, q.g_idx = L["qweight"].clone(), L["qzeros"].clone(), L["scales"].clone(), L["g_idx"].clone()
    if bias:
        q.bias = L["bias"].clone()
    q = q.to(DEV)
    q.post_init()
    mode = {"auto": O.ZERO_NOWRAP if (bits == 3 or act) else O.ZERO_WRAP, "wrap": O.ZERO_WRAP, "nowrap": O.ZERO_NOWRAP}[zero_mode]     # auto: qlinear_cuda_old's 3-bit branch and the act-order class do not wrap
    W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], bits, mode).to(DEV)
    return L, q, W


def _x(M, K, dtype, seed, hot=True):
    x = (torch.rand(M, K, generator=torch.Generator().manual_seed(seed)) - 0.5).to(dtype)
    ks = []
    if hot and M > 1:
        for r, k in zip(range(1, M), (K - 1, 0, 129 % K, 31 % K, (K // 2 + 1) % K, 64 % K, (K - 33) % K)):     # rows 4..7: the second A operand of the 5..8-row form
            x[r].zero_()
            x[r, k] = 1.0
            ks.append((r, k))
    return x.to(DEV), ks


def _assert_all(y, xd, W, bias, dtype, what):
    ref = xd.double() @ W.double()
    if bias is not None:
        ref = ref + bias.double()
    rtol, atol = TOL[dtype]
    scale = max(1e-6, float(ref.abs().max()))
    bad = (y.double() - ref).abs() > atol * scale + rtol * ref.abs()
    assert not bool(bad.any()), f"{what}: {int(bad.sum())}/{bad.numel()} outputs out of tolerance, first at {torch.nonzero(bad)[0].tolist()}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("K,N,gs", [(256, 96 + 32, 128), (160, 64, 32), (4096, 4096, 128), (1056, 48 + 16, 1056), (2112, 1040 - 16, 64)])
def test_prepack_decode_is_the_oracle_restatement(K, N, gs, dtype):
    import ctypes as C
    import numpy as np

    lib = _lib.load()
    for zm in ("auto", "nowrap"):
        L, q, W = _layer(K, N, gs, dtype, K + N, zero_mode=zm)
        mode = O.ZERO_WRAP if zm == "auto" else O.ZERO_NOWRAP
        tb, cb = C.c_size_t(0), C.c_size_t(0)
        _lib.check(lib.gptq_prepack_decode_bytes(C.byref(q._layer), C.byref(tb), C.byref(cb)))
        want_t = O.decode_copy_weights(L["qweight"])
        want_c = O.decode_copy_consts(L["qzeros"], L["scales"], mode)
        assert tb.value == want_t.numel() * 4 and cb.value == want_c.numel()
        assert q._qweight_tiled.numel() == tb.value and q._qconst_tiled.numel() == cb.value
        assert torch.equal(q._qweight_tiled.cpu().view(torch.int32).reshape(want_t.shape), want_t)
        assert torch.equal(q._qconst_tiled.cpu().reshape(want_c.shape), want_c)
        # lossless: the reference's integer unpack of the inverse is its unpack of the checkpoint tensor
        back = O.decode_copy_weights_inverse(q._qweight_tiled.cpu().view(torch.int32).reshape(want_t.shape), K)
        assert np.array_equal(O.unpack_weights(back, 4), O.unpack_weights(L["qweight"], 4))
    # argument errors surface as RuntimeError with the C ABI's message (the reference: TORCH_CHECK, exllama_ext.cpp:49-71)
    with pytest.raises(_lib.GptqError, match="in place"):
        _lib.check(lib.gptq_prepack_decode(C.byref(q._layer), q.qweight.data_ptr(), q._qconst_tiled.data_ptr(), None))
    L2 = O.random_quant_layer(256, 64, 4, 32, seed=1, dtype=torch.float32)
    q2 = QuantLinear(4, 32, 256, 64, False, weight_dtype=torch.float32)
    q2.qweight, q2.qzeros, q2.scales, q2.g_idx = L2["qweight"], L2["qzeros"], L2["scales"], L2["g_idx"]
    q2 = q2.to(DEV)
    q2.post_init()
    assert q2._qweight_tiled is None                                        # fp32 layers have no decode copy (2-bit fp16 / bf16 layers got one in round 6)
    with pytest.raises(_lib.GptqError, match="decode copy"):
        _lib.check(lib.gptq_prepack_decode_bytes(C.byref(q2._layer), C.byref(tb), C.byref(cb)))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("bits,K,N,gs", [(3, 256, 128, 128), (3, 160, 64, 32), (3, 4096, 1024, 128), (3, 1056, 64, 1056), (3, 2112, 1024, 64),
                                         (8, 256, 128, 128), (8, 160, 64, 16), (8, 4096, 1024, 128), (8, 1056, 64, 1056), (8, 2080, 1024, 32),
                                         (2, 256, 128, 128), (2, 160, 64, 32), (2, 4096, 1024, 128), (2, 1056, 64, 1056), (2, 2112, 1024, 64)])
def test_prepack_decode_3_and_8_bit_is_the_oracle_restatement(bits, K, N, gs, dtype):
    """The same for the 3-bit (three words per 32 k, no straddlers left) and 8-bit (16 k per lane, 2-byte zero-points) copies."""
    import ctypes as C

    lib = _lib.load()
    for zm in ("wrap", "nowrap"):
        L, q, W = _layer(K, N, gs, dtype, K + N, zero_mode=zm, bits=bits)
        mode = O.ZERO_WRAP if zm == "wrap" else O.ZERO_NOWRAP
        tb, cb = C.c_size_t(0), C.c_size_t(0)
        _lib.check(lib.gptq_prepack_decode_bytes(C.byref(q._layer), C.byref(tb), C.byref(cb)))
        want_t = O.decode_copy_weights(L["qweight"], bits)
        want_c = O.decode_copy_consts(L["qzeros"], L["scales"], mode, bits)
        assert tb.value == want_t.numel() * 4 and cb.value == want_c.numel()
        assert torch.equal(q._qweight_tiled.cpu().view(torch.int32).reshape(want_t.shape), want_t)
        assert torch.equal(q._qconst_tiled.cpu().reshape(want_c.shape), want_c)
        assert torch.equal(q.qweight.cpu(), L["qweight"]) and torch.equal(q.qzeros.cpu(), L["qzeros"])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("bits,K,N,gs", [(3, 1024, 512, 128), (3, 4160, 256, 32), (3, 2112, 1024, 64), (3, 96, 64, 32), (3, 1056, 64, 1056), (3, 4096, 4096, 128),
                                         (8, 1024, 512, 128), (8, 4128, 256, 16), (8, 2080, 1024, 32), (8, 96, 64, 32), (8, 1056, 64, 1056), (8, 4096, 4096, 128),
                                         (8, 512, 2048, 256), (3, 28672, 32, 128),
                                         (2, 1024, 512, 128), (2, 4160, 256, 32), (2, 2112, 1024, 64), (2, 96, 64, 32), (2, 1056, 64, 1056), (2, 4096, 4096, 128), (2, 28672, 32, 128)])
def test_tiled_decode_3_and_8_bit(bits, K, N, gs, dtype):
    """Every output of the 3- and 8-bit decode-copy kernels (BASELINE config 5's packings): default plan, forced geometries, K slices, 1..4 rows, both
    zero-point conventions (8-bit nowrap reaches z = 256: the 2-byte field), one-hot rows exact."""
    for zm in ("wrap", "nowrap"):
        L, q, W = _layer(K, N, gs, dtype, K + N + gs + bits, zero_mode=zm, bias=True, bits=bits)
        assert _lib.describe_plan(q._layer, 1)["kernel"] == "strips"
        for M in (1, 2, 3, 4):
            x, hot = _x(M, K, dtype, M)
            with torch.no_grad():
                y, y2 = q(x), q(x)
            assert torch.equal(y, y2)
            _assert_all(y, x, W, q.bias, dtype, f"tiled int{bits} default {K}x{N} g{gs} M={M} {dtype} {zm}")
            saved, q._layer.bias = q._layer.bias, None
            for t in (None, _tune(4, 4), _tune(16, 2), _tune(8, 2, 2), _tune(2, 4, 3)):
                with torch.no_grad():
                    y0 = q(x, tuning=t) if t is not None else q(x)
                for r, k in hot:
                    assert torch.equal(y0[r], W[k]), f"one-hot row {r} (k={k}) is not the oracle's W[k], int{bits} M={M} {K}x{N} g{gs}"
                _assert_all(y0, x, W, None, dtype, f"tiled int{bits} forced {K}x{N} g{gs} M={M}")
            q._layer.bias = saved


@pytest.mark.parametrize("bits", [2, 3, 8])
def test_tiled_multi_layer_launch_3_and_8_bit(bits):
    K = 2048
    widths = (512, 288, 64)
    made = [_layer(K, n, 128, torch.float16, 300 + n, bits=bits) for n in widths]
    layers = [m[1] for m in made]
    for M in (1, 4):
        x, hot = _x(M, K, torch.float16, M)
        with torch.no_grad():
            ys = forward_multi(layers, x, None)
        for i in range(3):
            for r, k in hot:
                assert torch.equal(ys[i][r], made[i][2][k])
            _assert_all(ys[i], x, made[i][2], None, torch.float16, f"tiled multi int{bits} layer {i} M={M}")


def test_side_copy_is_derived_and_checkpoint_untouched():
    L, q, W = _layer(512, 256, 128, torch.float16, 5)
    assert q._qweight_tiled is not None and q._qconst_tiled is not None and q._layer.tiled_cols == 16
    assert torch.equal(q.qweight.cpu(), L["qweight"]) and torch.equal(q.qzeros.cpu(), L["qzeros"]) and torch.equal(q.scales.cpu(), L["scales"])
    assert list(q.state_dict().keys()) == ["qweight", "qzeros", "scales", "g_idx"]
    assert _lib.describe_plan(q._layer, 1)["kernel"] == "strips"
    p = QuantLinear(4, 128, 512, 256, False)
    p.qweight, p.qzeros, p.scales, p.g_idx = q.qweight, q.qzeros, q.scales, q.g_idx
    p.post_init(tiled=False)
    assert p._qweight_tiled is None and _lib.describe_plan(p._layer, 1)["kernel"] != "strips"
    x, _ = _x(2, 512, torch.float16, 1)
    with torch.no_grad():
        _assert_all(p(x), x, W, None, torch.float16, "checkpoint-layout twin")
    with pytest.raises(_lib.GptqError, match="path = 8"):
        p(x, tuning=_tune())
    # act-order layers: the copy is made of the re-sequenced rows (the kernel gathers x through perm); the checkpoint tensors stay as they are
    La = O.random_quant_layer(512, 256, 4, 128, act_order=True, seed=2)
    a = QuantLinear(4, 128, 512, 256, False)
    a.qweight, a.qzeros, a.scales, a.g_idx = La["qweight"], La["qzeros"], La["scales"], La["g_idx"]
    a = a.to(DEV)
    a.post_init()
    assert a._qweight_tiled is not None and _lib.describe_plan(a._layer, 1)["kernel"] == "strips"
    assert torch.equal(a.qweight.cpu(), La["qweight"]) and torch.equal(a.g_idx.cpu(), La["g_idx"])
    # the copy == the oracle's restatement applied to the oracle's re-sequenced rows (rows of w in the stable group order, re-packed)
    import numpy as np
    perm = O.sequential_permutation(La["g_idx"])
    seq = torch.from_numpy(O.pack_rows(O.unpack_weights(La["qweight"], 4)[perm], 4))
    want = O.decode_copy_weights(seq)
    assert torch.equal(a._qweight_tiled.cpu().view(torch.int32).reshape(want.shape), want)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("K,N,gs", [(1024, 512, 128), (4160, 272 - 16, 32), (2112, 1040 - 16, 64), (512, 2048, 256), (96, 64, 32), (3200, 32, 128), (1056, 64, 1056), (28672, 32, 128)])
def test_tiled_decode_default_plan(K, N, gs, dtype):
    """Default plan at shapes with whole and ragged chunks (K / 8 not a multiple of 16), strips that do not fill a wave count evenly,
    one strip only, and every supported group size."""
    for zm in ("auto", "nowrap"):
        L, q, W = _layer(K, N, gs, dtype, K + N + gs, zero_mode=zm, bias=True)
        assert _lib.describe_plan(q._layer, 1)["kernel"] == "strips"
        for M in (1, 2, 3, 4):
            x, hot = _x(M, K, dtype, M)
            with torch.no_grad():
                y, y2 = q(x), q(x)
            assert torch.equal(y, y2)
            _assert_all(y, x, W, q.bias, dtype, f"tiled default {K}x{N} g{gs} M={M} {dtype} {zm}")
            if hot:                              # exactness without the bias (it is added on fp32 before the one rounding; the reference rounds twice)
                saved, q._layer.bias = q._layer.bias, None
                with torch.no_grad():
                    y0 = q(x)
                q._layer.bias = saved
                for r, k in hot:
                    assert torch.equal(y0[r], W[k]), f"one-hot row {r} (k={k}) is not the oracle's W[k], M={M} {K}x{N} g{gs}"


@pytest.mark.parametrize("waves,u", [(16, 2), (8, 2), (8, 4), (4, 2), (4, 4), (2, 4), (1, 2), (3, 4)])
def test_tiled_decode_forced_geometries(waves, u):
    """Every (waves, chunks per wave) the planner or a sweep can ask for, incl. workgroups that walk their strip in several passes and ones larger
    than the strip.  (16 waves x 4 chunks is no longer compiled for single-strip workgroups -- the next test.)"""
    for (K, N, gs), dtype in (((2048, 256, 128), torch.float16), ((4160, 64, 64), torch.bfloat16)):
        L, q, W = _layer(K, N, gs, dtype, waves * 31 + u)
        for M in (1, 4):
            x, hot = _x(M, K, dtype, M + u)
            with torch.no_grad():
                y = q(x, tuning=_tune(waves, u))
                y2 = q(x, tuning=_tune(waves, u))
            assert torch.equal(y, y2)
            for r, k in hot:
                assert torch.equal(y[r], W[k])
            _assert_all(y, x, W, None, dtype, f"tiled waves={waves} u={u} {K}x{N} M={M}")


def test_tiled_decode_geometry_outside_its_compilation_is_refused():
    """Round 6: every decode-copy form is compiled ONCE -- 2 chunks in flight for workgroups of up to 16 waves, 4 chunks for up to 8 (the planner's
    geometries; the two compilations per form of rounds 4-5 differed by a register or two).  A forced 16 x 4 is refused by the planner (status, message,
    nothing launched); the default plan of the same layer runs."""
    L, q, W = _layer(2048, 256, 128, torch.float16, 5)
    x, _ = _x(2, 2048, torch.float16, 3)
    with pytest.raises(_lib.GptqError):
        q(x, tuning=_tune(16, 4))
    with torch.no_grad():
        _assert_all(q(x), x, W, None, torch.float16, "default after a refused geometry")


@pytest.mark.parametrize("ks", [2, 3, 4, 8])
def test_tiled_decode_k_slices(ks):
    """K slices combined inside the launch through granules (narrow layers: TP shards): forced, and the planner's own on a 64-strip layer."""
    for (K, N), dtype in (((8192, 256, ), torch.float16), ((4160, 128), torch.bfloat16)):
        L, q, W = _layer(K, N, 128, dtype, ks)
        for M in (1, 3):
            x, hot = _x(M, K, dtype, M)
            for waves, u in ((8, 2), (16, 2), (4, 4)):
                with torch.no_grad():
                    y = q(x, tuning=_tune(waves, u, ks))
                    y2 = q(x, tuning=_tune(waves, u, ks))
                assert torch.equal(y, y2), "K-split combine is not bit-reproducible"
                for r, k in hot:
                    assert torch.equal(y[r], W[k])
                _assert_all(y, x, W, None, dtype, f"tiled ksplit={ks} {K}x{N} M={M}")
    L, q, W = _layer(8192, 1024, 128, torch.float16, 77)                    # Llama-2-70B attention shard at TP = 8: no K slices since late round 6 (the hop costs more than the idle CUs)
    plan = _lib.describe_plan(q._layer, 1)
    assert plan["kernel"] == "strips" and plan["ksplit"] == 1 and (plan["waves"], plan["u"]) == (8, 4), plan
    x, _ = _x(1, 8192, torch.float16, 3)
    with torch.no_grad():
        _assert_all(q(x), x, W, None, torch.float16, "tiled 8192x1024 default")
    L, q, W = _layer(28672, 1024, 128, torch.float16, 78)                   # the down-projection shard: slices of 7168 k (deep enough to pay for the hop)
    plan = _lib.describe_plan(q._layer, 1)
    assert plan["kernel"] == "strips" and plan["ksplit"] == 4, plan
    x, _ = _x(1, 28672, torch.float16, 4)
    with torch.no_grad():
        _assert_all(q(x), x, W, None, torch.float16, "tiled 28672x1024 default (K slices)")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_tiled_multi_layer_launch(dtype):
    """2..4 layers of DIFFERENT widths that read one x, in one launch (gptq_forward_multi): every output of every layer."""
    K = 2048
    widths = (512, 288, 1024, 64)
    made = [_layer(K, n, 128, dtype, 300 + n) for n in widths]
    for n_l in (2, 3, 4):
        layers = [m[1] for m in made[:n_l]]
        for M in (1, 2, 4):
            x, hot = _x(M, K, dtype, M)
            for t in (None, _tune(4, 4), _tune(8, 2), _tune(16, 2, 2)):
                with torch.no_grad():
                    ys = forward_multi(layers, x, t)
                    ys2 = forward_multi(layers, x, t)
                for i in range(n_l):
                    assert torch.equal(ys[i], ys2[i])
                    for r, k in hot:
                        assert torch.equal(ys[i][r], made[i][2][k])
                    _assert_all(ys[i], x, made[i][2], None, dtype, f"tiled multi n={n_l} layer {i} M={M}")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("bits,K,N,gs", [(4, 1024, 512, 128), (4, 4096, 4096, 128), (4, 4160, 256, 32), (4, 2112, 1024, 64), (4, 96, 64, 32), (4, 11008, 256, 128),
                                         (4, 28672, 32, 128), (3, 2048, 512, 128), (3, 4096, 1024, 32), (8, 2048, 512, 128), (8, 4128, 256, 16),
                                         (2, 2048, 512, 128), (2, 4160, 256, 32)])
def test_tiled_decode_act_order(bits, K, N, gs, dtype):
    """Act-order (desc_act=True) layers through the decode-copy kernel: the copy holds the re-sequenced rows, the workgroup gathers x[perm[i]] while it stages
    x.  Every output against x (fp64) @ W_oracle (fp64) with the act-order class's zero convention, one-hot rows exact (a one at ORIGINAL position k must
    return W[k]), default plan + forced geometries + K slices, 1..4 rows, long K (several gather passes per thread)."""
    L, q, W = _layer(K, N, gs, dtype, K + N + gs + bits, bias=True, bits=bits, act=True)
    assert q._qweight_tiled is not None and _lib.describe_plan(q._layer, 1)["kernel"] == "strips"
    for M in (1, 2, 3, 4):
        x, hot = _x(M, K, dtype, M)
        with torch.no_grad():
            y, y2 = q(x), q(x)
        assert torch.equal(y, y2)
        _assert_all(y, x, W, q.bias, dtype, f"tiled act-order int{bits} default {K}x{N} g{gs} M={M} {dtype}")
        saved, q._layer.bias = q._layer.bias, None
        for t in (None, _tune(4, 4), _tune(16, 2), _tune(8, 2, 2), _tune(2, 4, 3)):
            if t is not None:                     # the raw rows of x (whole K) may not fit the LDS at this M: such calls keep the in-kernel-gather kernels
                try:
                    if _lib.describe_plan(q._layer, M, t).get("kernel") != "strips":
                        continue
                except _lib.GptqError:
                    continue
            with torch.no_grad():
                y0 = q(x, tuning=t) if t is not None else q(x)
            for r, k in hot:
                assert torch.equal(y0[r], W[k]), f"one-hot row {r} (k={k}) is not the oracle's W[k], act-order int{bits} M={M} {K}x{N} g{gs}"
            _assert_all(y0, x, W, None, dtype, f"tiled act-order int{bits} forced {K}x{N} g{gs} M={M}")
        q._layer.bias = saved


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("bits,K,N,gs,act", [(4, 1024, 512, 128, False), (4, 4160, 256, 32, False), (4, 11008, 64, 128, False), (4, 96, 64, 32, False), (3, 2112, 256, 64, False),
                                             (8, 2048, 512, 128, False), (8, 4128, 64, 16, False), (4, 2048, 512, 128, True), (3, 4096, 256, 32, True), (8, 2048, 256, 128, True)])
def test_tiled_decode_5_to_8_rows(bits, K, N, gs, act, dtype):
    """5..8 rows of x on the decode copy (MT = 8: rows 4..7 are a second A operand, two matrix-core steps per decoded pair; asked for with tuning.path = 8):
    every output against x (fp64) @ W_oracle (fp64), one-hot rows in BOTH halves exact, default and forced geometries, K slices (long K: the planner's own --
    eight staged rows of x do not fit one workgroup's LDS), bit-reproducible."""
    L, q, W = _layer(K, N, gs, dtype, K + N + gs + bits + 5, bits=bits, act=act)
    for M in (5, 6, 8):
        x, hot = _x(M, K, dtype, M)
        for t in (_tune(), _tune(16, 2), _tune(4, 4), _tune(8, 2, 2)):
            try:
                if _lib.describe_plan(q._layer, M, t).get("kernel") != "strips":
                    continue
            except _lib.GptqError:                # (act-order: the raw rows of x, whole K, may not fit the LDS)
                continue
            with torch.no_grad():
                y, y2 = q(x, tuning=t), q(x, tuning=t)
            assert torch.equal(y, y2)
            for r, k in hot:
                assert torch.equal(y[r], W[k]), f"one-hot row {r} (k={k}) is not the oracle's W[k], int{bits} M={M} {K}x{N} g{gs} act={act}"
            _assert_all(y, x, W, None, dtype, f"tiled int{bits} {K}x{N} g{gs} M={M} act={act} waves={t.waves} u={t.reserved[0]} ks={t.ksplit}")
    if not act and bits == 4 and K == 1024:       # several layers in one launch, 8 rows
        L2, q2, W2 = _layer(K, 288, gs, dtype, 99, bits=bits)
        x, hot = _x(8, K, dtype, 8)
        with torch.no_grad():
            ys = forward_multi([q, q2], x, _tune())
        for y, Wi in zip(ys, (W, W2)):
            for r, k in hot:
                assert torch.equal(y[r], Wi[k])
            _assert_all(y, x, Wi, None, dtype, "tiled multi, 8 rows")


def test_tiled_multi_layer_launch_act_order():
    """q | k | v-like act-order layers in ONE launch, each with its OWN perm (and once sharing one g_idx): every output, every layer."""
    K = 2048
    made = [_layer(K, n, 128, torch.float16, 700 + n, act=True) for n in (512, 288, 64)]
    layers = [m[1] for m in made]
    for M in (1, 4):
        x, hot = _x(M, K, torch.float16, M)
        with torch.no_grad():
            ys = forward_multi(layers, x, None)
            ys2 = forward_multi(layers, x, None)
        for i in range(3):
            assert torch.equal(ys[i], ys2[i])
            for r, k in hot:
                assert torch.equal(ys[i][r], made[i][2][k])
            _assert_all(ys[i], x, made[i][2], None, torch.float16, f"tiled multi act-order layer {i} M={M}")
    # a plain and an act-order layer do not share a launch (different staging): layer by layer, same results
    Lp, qp, Wp = _layer(K, 128, 128, torch.float16, 5)
    x, _ = _x(2, K, torch.float16, 9)
    with torch.no_grad():
        ym = forward_multi([qp, layers[0]], x, None)
    _assert_all(ym[0], x, Wp, None, torch.float16, "mixed group, plain layer")
    _assert_all(ym[1], x, made[0][2], None, torch.float16, "mixed group, act-order layer")


def test_sticky_exchange_error_is_read_periodically():
    """A bounded in-launch wait that gives up raises the workspace's sticky error word (the kernels never hang); QuantLinear reads it every
    EXCHANGE_CHECK_EVERY workspace-taking calls instead of leaving it to the caller (round-3 advice)."""
    from autogptq_amd import qlinear_mi355x as qm
    L, q, W = _layer(28672, 256, 128, torch.float16, 3)                   # 16 strips of a 28672-deep layer: K slices (of 7168 k), i.e. a workspace and an exchange
    assert _lib.describe_plan(q._layer, 1)["ksplit"] >= 2
    x, _ = _x(1, 28672, torch.float16, 1)
    saved = QuantLinear.EXCHANGE_CHECK_EVERY
    QuantLinear.EXCHANGE_CHECK_EVERY = 1
    try:
        with torch.no_grad():
            y = q(x)                                                      # checked: clean
        assert not qm.exchange_error(DEV)
        ws = qm._WORKSPACE[(torch.cuda.current_device(), int(torch.cuda.current_stream().cuda_stream))][0]
        tail = ws[_lib.WS_HEADER_BYTES - 64:_lib.WS_HEADER_BYTES].view(torch.int32)
        tail[2] = 1                                                       # what a wait that gave up leaves behind
        with pytest.raises(RuntimeError, match="bounded wait"):
            with torch.no_grad():
                q(x)
        assert int(tail[2].item()) == 0                                   # reported once, then cleared (round-4 advice): one timeout, one exception
        with torch.no_grad():
            assert torch.equal(q(x), y)
        tail[2] = 1                                                       # the multi-layer entry point reads the same word (it takes the same workspace)
        with pytest.raises(RuntimeError, match="bounded wait"):
            with torch.no_grad():
                qm.forward_multi([q], x)
        assert int(tail[2].item()) == 0
    finally:
        QuantLinear.EXCHANGE_CHECK_EVERY = saved


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("K,I,gs,bits", [(1024, 704, 128, 4), (4096, 11008, 128, 4), (2112, 96, 64, 4), (1024, 1408, 32, 3), (1024, 1408, 32, 8), (160, 32, 32, 4)],
                         ids=["1024x704", "llama7b-mlp", "ragged-k", "int3", "int8", "tiny"])
def test_gate_up_silu_mul_is_the_epilogue_of_the_decode_copy_kernel(K, I, gs, bits, dtype):
    """Round 5: a [gate | up] layer (epilogue='silu_mul', the reference's fused MLP: auto_gptq/nn_modules/fused_llama_mlp.py:131-306) carries its decode copy and
    its decode rows run the PAIR form of gemv_tiled_kernel (csrc/gemv_tiled_pair.hip): plan pinned, EVERY output against silu(x @ W_gate + b) * (x @ W_up + b)
    formed in fp64 from the oracle's weights, repeated calls bit-identical, checkpoint tensors untouched; forced geometries too."""
    from autogptq_amd.fused import fuse_gate_up
    Lg = O.random_quant_layer(K, I, bits, gs, dtype=dtype, seed=K + I, bias=True)
    Lu = O.random_quant_layer(K, I, bits, gs, dtype=dtype, seed=K + I + 1, bias=True)
    for L in (Lg, Lu):
        L["scales"] = (L["scales"].float() * (8 if K < 2048 else 4)).to(dtype)      # gate pre-activations of order 1: SiLU off its linear part
    mods = []
    for L in (Lg, Lu):
        m = QuantLinear(bits, gs, K, I, True, weight_dtype=dtype)
        m.qweight, m.qzeros, m.scales, m.g_idx, m.bias = L["qweight"].clone(), L["qzeros"].clone(), L["scales"].clone(), L["g_idx"].clone(), L["bias"].clone()
        mods.append(m)
    fused = fuse_gate_up(*mods).to(DEV)
    q = next(m for m in fused.modules() if isinstance(m, QuantLinear))
    q.post_init()
    assert q._qweight_tiled is not None, "a plain [gate | up] layer gets a decode copy"
    before = {k: v.clone() for k, v in q.state_dict().items()}
    mode = O.reference_zero_mode(False, bits)
    Wg = O.dequantize(Lg["qweight"], Lg["qzeros"], Lg["scales"], Lg["g_idx"], bits, mode).to(DEV).double()
    Wu = O.dequantize(Lu["qweight"], Lu["qzeros"], Lu["scales"], Lu["g_idx"], bits, mode).to(DEV).double()
    bg, bu = Lg["bias"].to(DEV).double(), Lu["bias"].to(DEV).double()
    rtol, atol = {torch.float16: (2e-3, 2e-3), torch.bfloat16: (1.6e-2, 1.6e-2)}[dtype]       # one rounding of the product of two sums
    for M in (1, 2, 3, 4):
        x, _ = _x(M, K, dtype, M, hot=False)
        plans = [None] + ([_tune(16, 2), _tune(8, 4), _tune(4, 4), _tune(2, 2)] if M in (1, 4) else [])
        for t in plans:
            d = _lib.describe_plan(q._layer, M, t)
            assert (d["kernel"], int(d["pair"]), d["epilogue"]) == ("strips", 1, "fused") and int(d["strips"]) == I // 16, d
            with torch.no_grad():
                y, y2 = q(x, tuning=t), q(x, tuning=t)
            assert tuple(y.shape) == (M, I) and torch.equal(y, y2)
            ref = torch.nn.functional.silu(x.double() @ Wg + bg) * (x.double() @ Wu + bu)
            scale = float(ref.abs().max())
            bad = (y.double() - ref).abs() > atol * scale + rtol * ref.abs()
            assert not bool(bad.any()), f"M={M} {d}: {int(bad.sum())}/{bad.numel()} outputs out of tolerance, first {torch.nonzero(bad)[0].tolist()}"
    assert float((x.double() @ Wg + bg).abs().max()) > 0.5
    for k, v in q.state_dict().items():
        assert torch.equal(v, before[k]), k
    with torch.no_grad():                                                           # 5+ rows: the staged path as before, on the same layer
        x8, _ = _x(8, K, dtype, 8, hot=False)
        y8 = q(x8)
    ref8 = torch.nn.functional.silu(x8.double() @ Wg + bg) * (x8.double() @ Wu + bu)
    assert not bool(((y8.double() - ref8).abs() > 3 * atol * float(ref8.abs().max()) + 3 * rtol * ref8.abs()).any())


@pytest.mark.parametrize("bits,gs", [(4, 128), (8, 32), (3, 64)])
def test_fused_qkv_g_idx_of_three_activation_orders(bits, gs):
    """``len(g_idx) == 3 K``: the reference's fused q/k/v module of act-order projections (fused_llama_attn.py:186 concatenates the three g_idx;
    qlinear_cuda.py:300-312 dequantises column block i with g_idx[i K : (i + 1) K]).  One module holding the concatenated checkpoint tensors must give, for
    every row count, the outputs of the three projections run on their own -- every output against the oracle's per-block weights -- and dequantize() their
    concatenation, with the state_dict left as loaded; a g_idx length that is no multiple of K is still refused."""
    K = 512
    Ls = [O.random_quant_layer(K, K, bits, gs, act_order=True, seed=70 + i, bias=True) for i in range(3)]
    f = QuantLinear(bits, gs, K, 3 * K, True)
    f.qweight = torch.cat([L["qweight"] for L in Ls], dim=1)
    f.qzeros = torch.cat([L["qzeros"] for L in Ls], dim=1)
    f.scales = torch.cat([L["scales"] for L in Ls], dim=1)
    f.g_idx = torch.cat([L["g_idx"] for L in Ls], dim=0)
    f.bias = torch.cat([L["bias"] for L in Ls], dim=0)
    f = f.to(DEV)
    before = {k: v.clone() for k, v in f.state_dict().items()}
    mode = O.reference_zero_mode(True, bits)
    W = torch.cat([O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], bits, mode) for L in Ls], dim=1).to(DEV)
    bias = torch.cat([L["bias"] for L in Ls]).to(DEV)
    for M in (1, 3, 8, 40, 300):
        x, _ = _x(M, K, torch.float16, M, hot=False)
        with torch.no_grad():
            y, y2 = f(x), f(x)
        assert tuple(y.shape) == (M, 3 * K) and torch.equal(y, y2)
        _assert_all(y, x, W, bias, torch.float16, f"fused q|k|v, three activation orders, M={M}")
    with torch.no_grad():
        assert torch.equal(f.dequantize(), W)
    assert set(f.state_dict()) == set(before) and all(torch.equal(f.state_dict()[k], before[k]) for k in before)
    bad = QuantLinear(bits, gs, K, 3 * K, False)
    bad.g_idx = torch.zeros(K + 32, dtype=torch.int32)
    with pytest.raises(NotImplementedError, match="n \\* infeatures"):
        bad.to(DEV).post_init()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("nstr", [2, 4])
def test_several_strips_per_workgroup_behind_one_staged_x(nstr, dtype):
    """Round 5 (gemv_tiled_multi.hip, XM = 5 / 6): FOUR / TWO adjacent strips of a layer per workgroup share ONE staged x; the waves split into one group per
    strip and the cross-wave sum is per strip.  Forced (tuning.path = 8, reserved[GPTQ_LAB_OPT] = strips per workgroup) at 3..8 rows on a single layer and on
    a three-layer launch; EVERY output against the oracle, one-hot rows exact, bit-reproducible, and the planner's own choice at 5..8 rows is that form."""
    LAB = _lib.LAB
    K = 1024
    made = [_layer(K, N, 128, dtype, 300 + i, bias=True) for i, N in enumerate((6400, 1024, 1024))]
    layers = [m[1] for m in made]
    for M in ((1, 2, 3, 4, 5, 8) if nstr == 2 else (3, 4, 5, 8)):                # round 6: two strips per workgroup also at 1 - 2 rows (4-bit layers)
        x, ks = _x(M, K, dtype, M)
        t = _tune()
        t.reserved[LAB.OPT] = nstr
        d = _lib.describe_plan(layers[0]._layer, M, t)
        assert d["kernel"] == "strips" and int(d["strips"]) == 6400 // 16 // nstr, d
        with torch.no_grad():
            y, y2 = layers[0](x, tuning=t), layers[0](x, tuning=t)
            ym = forward_multi(layers, x, t)
        assert torch.equal(y, y2)
        _assert_all(y, x, made[0][2], made[0][0]["bias"].to(DEV), dtype, f"{nstr} strips per workgroup, M={M}")
        for (L, q, W), yy in zip(made, ym):
            _assert_all(yy, x, W, L["bias"].to(DEV), dtype, f"{nstr} strips per workgroup, three layers in one launch, M={M}, N={q.outfeatures}")
        assert torch.equal(ym[0], y)
        saved = [q._layer.bias for q in layers]
        for q in layers:
            q._layer.bias = None
        with torch.no_grad():
            yh = layers[0](x, tuning=t)
        for q, b in zip(layers, saved):
            q._layer.bias = b
        for r, k in ks:
            assert torch.equal(yh[r], made[0][2][k]), f"one-hot row {r} -> k={k} (M={M})"
    t8 = _tune()
    d = _lib.describe_plan(layers[0]._layer, 8, t8)                 # path = 8 without a count: the planner's own (two strips per workgroup only from 1024 strips)
    assert int(d["strips"]) == 6400 // 16, d


@pytest.mark.parametrize("bits,K,N,gs", [(4, 4096, 4096, 128), (4, 160, 64, 32), (8, 2112, 1024, 64), (3, 1056, 64, 1056), (3, 4096, 256, 32), (8, 96, 32, 32),
                                         (2, 4096, 1024, 128), (2, 160, 64, 32)])
def test_unprepack_decode_is_the_exact_inverse(bits, K, N, gs):
    """gptq_unprepack_decode(gptq_prepack_decode(qweight)) == qweight, bit for bit, for every packing with a decode copy (ragged last chunks, the 3-bit
    straddlers); at 4 bits also the oracle's own inverse (decode_copy_weights_inverse)."""
    lib = _lib.load()
    L, q, W = _layer(K, N, gs, torch.float16, K + N + bits, bits=bits)
    assert q._qweight_tiled is not None
    back = torch.empty_like(q.qweight)
    _lib.check(lib.gptq_unprepack_decode(q._qweight_tiled.data_ptr(), K, N, bits, back.data_ptr(), _lib.current_stream_handle(torch.device(DEV))))
    torch.cuda.synchronize()
    assert torch.equal(back.cpu(), L["qweight"]), "the inverse does not restore the checkpoint rows"
    if bits == 4:
        chunks = -(-K // 128)
        t = q._qweight_tiled.view(torch.int32).reshape(N // 16, chunks, 4, 16, 4).cpu()
        assert torch.equal(O.decode_copy_weights_inverse(t, K), L["qweight"])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
def test_release_checkpoint_layout_keeps_one_copy_on_the_device(dtype):
    """post_init(release_checkpoint_layout=True): qweight of a plain layer moves to pinned host memory (state_dict unchanged, bit for bit), decode and prefill
    rows run from the decode copy alone and the row counts in between rebuild the packed rows into the shared scratch -- every row count gives the SAME bits
    as the layer that keeps both layouts; two released layers of different shapes interleave on one scratch; .to() brings the rows back."""
    from autogptq_amd import qlinear_mi355x as qm
    shapes = [(1024, 2048, 128), (2048, 512, 64)]
    pairs = []
    for i, (K, N, gs) in enumerate(shapes):
        L = O.random_quant_layer(K, N, 4, gs, dtype=dtype, seed=500 + i, bias=True)
        mods = []
        for rel in (False, True):
            m = QuantLinear(4, gs, K, N, True, weight_dtype=dtype)
            m.qweight, m.qzeros, m.scales, m.g_idx, m.bias = L["qweight"].clone(), L["qzeros"].clone(), L["scales"].clone(), L["g_idx"].clone(), L["bias"].clone()
            m = m.to(DEV)
            m.post_init(release_checkpoint_layout=rel)
            mods.append(m)
        keep, rel = mods
        assert rel._released and rel.qweight.device.type == "cpu" and rel.qweight.is_pinned() and keep.qweight.device.type == "cuda"
        assert torch.equal(rel.state_dict()["qweight"].cpu(), L["qweight"]) and set(rel.state_dict()) == set(keep.state_dict())
        pairs.append((L, keep, rel))
    for M in (1, 4, 8, 16, 64, 200, 1024):
        for (L, keep, rel) in pairs:                            # interleaved: the second layer's rebuild overwrites the scratch the first one used
            x, _ = _x(M, L["K"], dtype, M, hot=False)
            with torch.no_grad():
                assert torch.equal(rel(x), keep(x)), f"released layer differs at M={M}"
    with torch.no_grad():
        for (L, keep, rel) in pairs:
            assert torch.equal(rel.dequantize(), keep.dequantize())
            x, _ = _x(16, L["K"], dtype, 3, hot=False)
            ym = qm.forward_multi([rel], x)[0]
            assert torch.equal(ym, keep(x))
    L, keep, rel = pairs[0]
    rel2 = rel.to(DEV)                                          # the rows come back with the module
    assert rel2.qweight.device.type == "cuda" and torch.equal(rel2.qweight.cpu(), L["qweight"])
    x, _ = _x(16, L["K"], dtype, 5, hot=False)
    with torch.no_grad():
        assert torch.equal(rel2(x), keep(x))


def test_autogptq_post_init_on_a_model_with_a_fused_qkv_g_idx_layer_and_with_released_rows():
    """autogptq_post_init (the model-level entry point, auto_gptq/modeling/_utils.py:380-513 in the reference): (i) a module whose g_idx has n * K entries (the
    reference's fused q/k/v of act-order projections) is initialised through its n internal column blocks -- their workspace needs are counted, nothing is
    dereferenced through the module's own (absent) layer struct; (ii) release_checkpoint_layout=True moves qweight to host memory and the scratch is still
    sized on the LAYER'S GPU, so the forward calls that follow allocate nothing (the scratch tensor of the stream is the one post_init reserved)."""
    import torch.nn as nn
    from autogptq_amd import qlinear_mi355x as qm
    from autogptq_amd.model_utils import autogptq_post_init
    K = 512
    Ls = [O.random_quant_layer(K, K, 4, 128, act_order=True, seed=170 + i, bias=True) for i in range(3)]
    f = QuantLinear(4, 128, K, 3 * K, True)
    f.qweight = torch.cat([L["qweight"] for L in Ls], dim=1)
    f.qzeros = torch.cat([L["qzeros"] for L in Ls], dim=1)
    f.scales = torch.cat([L["scales"] for L in Ls], dim=1)
    f.g_idx = torch.cat([L["g_idx"] for L in Ls], dim=0)
    f.bias = torch.cat([L["bias"] for L in Ls], dim=0)
    Lp = O.random_quant_layer(K, 1024, 4, 128, seed=180, bias=True)
    p = QuantLinear(4, 128, K, 1024, True)
    p.qweight, p.qzeros, p.scales, p.g_idx, p.bias = Lp["qweight"], Lp["qzeros"], Lp["scales"], Lp["g_idx"], Lp["bias"]

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.qkv, self.proj = f, p
    model = Block().to(DEV)
    model = autogptq_post_init(model, use_act_order=True, max_input_length=300, release_checkpoint_layout=True)
    assert model.qkv._parts is not None and len(model.qkv._parts) == 3
    assert model.proj._released and model.proj.qweight.device.type == "cpu"
    idx = torch.device(DEV).index or 0
    key = (idx, int(torch.cuda.current_stream(DEV).cuda_stream))
    assert key in qm._WORKSPACE and all(k[0] == idx for k in qm._WORKSPACE), "the scratch was sized on another device than the layers'"
    reserved = qm._WORKSPACE[key][0]
    mode = O.reference_zero_mode(True, 4)
    W = torch.cat([O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 4, mode) for L in Ls], dim=1).to(DEV)
    bias = torch.cat([L["bias"] for L in Ls]).to(DEV)
    Wp = O.dequantize(Lp["qweight"], Lp["qzeros"], Lp["scales"], Lp["g_idx"], 4, O.reference_zero_mode(False, 4)).to(DEV)
    for M in (1, 4, 40, 200, 300):
        x, _ = _x(M, K, torch.float16, M, hot=False)
        with torch.no_grad():
            y, yp = model.qkv(x), model.proj(x)
        _assert_all(y, x, W, bias, torch.float16, f"fused q|k|v after autogptq_post_init, M={M}")
        _assert_all(yp, x, Wp, Lp["bias"].to(DEV), torch.float16, f"released layer after autogptq_post_init, M={M}")
        assert qm._WORKSPACE[key][0] is reserved, f"forward at M={M} replaced the scratch autogptq_post_init reserved (an allocation on the forward path)"
