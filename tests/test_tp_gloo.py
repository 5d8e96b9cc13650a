"""CPU (-m "not gpu"), 2 processes over gloo: the out_features tensor-parallel split (shard_packed) and the
single all-gather that reassembles [M, N] (ColumnParallelQuantLinear).  The rank-local matmul is played by the
oracle here (this is a test of the sharding + exchange logic; the HIP kernels are covered by the -m gpu tests,
whose column-slice property test proves a shard computes exactly the columns it owns)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from autogptq_amd.tensor_parallel import ColumnParallelQuantLinear, shard_bounds, shard_packed
from oracle import gptq_oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, bits, gs, K, N, M, act, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        L = O.random_quant_layer(K, N, bits, gs, act_order=act, seed=3, bias=True)     # same on every rank
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(5)) - 0.5).float()
        mode = O.reference_zero_mode(act, bits)
        qw, qz, sc, b = shard_packed(L["qweight"], L["qzeros"], L["scales"].float(), L["bias"].float(), bits, rank, world)

        def local(xx):
            return O.forward(xx, qw, qz, sc, L["g_idx"], b, bits, mode)

        mod = ColumnParallelQuantLinear(local, N)
        y = mod(x)
        ref = O.forward(x, L["qweight"], L["qzeros"], L["scales"].float(), L["g_idx"], L["bias"].float(), bits, mode)
        ok = tuple(y.shape) == (M, N) and torch.allclose(y, ref, rtol=1e-5, atol=1e-6)
        y3 = mod(x.reshape(1, M, K))                      # leading dims preserved
        ok = ok and tuple(y3.shape) == (1, M, N) and torch.equal(y3[0], y)
        # the direct peer-store exchange moves device buffers through IPC mappings: host tensors are refused, on every rank, before
        # anything collective happens (the GPU side is tests/test_peer_exchange.py)
        try:
            ColumnParallelQuantLinear(local, N, exchange="peer_store")(x)
            ok = False
        except RuntimeError as e:
            ok = ok and "peer_store" in str(e)
        q.put((rank, bool(ok), float((y - ref).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bits,gs,K,N,M,act", [(4, 128, 256, 256, 1, False), (4, 32, 128, 192, 3, True), (3, 32, 128, 128, 2, False),
                                              (8, 64, 128, 64, 2, True)])
def test_column_parallel_two_ranks_gloo(bits, gs, K, N, M, act):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bits, gs, K, N, M, act, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_shard_bounds_rules():
    assert shard_bounds(4096, 3, 8) == (1536, 2048)
    assert shard_bounds(11008, 7, 8) == (9632, 11008)       # 1376 columns per rank
    assert shard_bounds(28672, 0, 8) == (0, 3584)
    with pytest.raises(ValueError):
        shard_bounds(4096 + 32, 0, 8)                        # not a multiple of 32 columns per rank
    with pytest.raises(ValueError):
        shard_bounds(96, 0, 8)


def _worker_row(rank, world, port, bits, gs, K, N, M, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autogptq_amd.tensor_parallel import RowParallelQuantLinear, shard_packed_rows
        L = O.random_quant_layer(K, N, bits, gs, seed=4, bias=True)
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(6)) - 0.5).float()
        mode = O.reference_zero_mode(False, bits)
        qw, qz, sc, (k0, k1) = shard_packed_rows(L["qweight"], L["qzeros"], L["scales"].float(), bits, gs, rank, world)

        def local(xx):                                   # the rank's [M, K/T] x [K/T, N] partial product (oracle as stand-in)
            return O.forward(xx, qw, qz, sc, None, None, bits, mode)

        mod = RowParallelQuantLinear(local, (k0, k1), bias=L["bias"].float(), input_is_parallel=False)
        y = mod(x)
        ref = O.forward(x, L["qweight"], L["qzeros"], L["scales"].float(), L["g_idx"], L["bias"].float(), bits, mode)
        q.put((rank, bool(torch.allclose(y, ref, rtol=1e-4, atol=1e-5)), float((y - ref).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bits,gs,K,N,M", [(4, 128, 512, 64, 2), (3, 32, 256, 96, 1), (8, 64, 256, 32, 3)])
def test_row_parallel_two_ranks_gloo(bits, gs, K, N, M):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_row, args=(r, world, port, bits, gs, K, N, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def _worker_row_threshold(rank, world, port, q):
    """The two reductions of RowParallelQuantLinear on the SAME fp16 partial sums: threshold below the output size (one pass in fp16, every partial rounded
    before the sum) against threshold None (fp32 sum, one rounding)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autogptq_amd.tensor_parallel import RowParallelQuantLinear, shard_packed_rows
        bits, gs, K, N, M = 4, 128, 1024, 256, 8
        L = O.random_quant_layer(K, N, bits, gs, seed=9, bias=True)
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(3)) - 0.5).half()
        mode = O.reference_zero_mode(False, bits)
        qw, qz, sc, (k0, k1) = shard_packed_rows(L["qweight"], L["qzeros"], L["scales"], bits, gs, rank, world)

        def local(xx):                                   # fp16 partial product, as the HIP shard returns it
            return O.forward(xx.float(), qw, qz, sc.float(), None, None, bits, mode).half()

        lo = RowParallelQuantLinear(local, (k0, k1), bias=L["bias"], input_is_parallel=False, fp32_reduce_max_elems=M * N - 1)
        hi = RowParallelQuantLinear(local, (k0, k1), bias=L["bias"], input_is_parallel=False, fp32_reduce_max_elems=None)
        assert lo.fp32_reduce_max_elems == M * N - 1 and hi.fp32_reduce_max_elems is None
        y_lo, y_hi = lo(x), hi(x)
        ref = O.forward(x.float(), L["qweight"], L["qzeros"], L["scales"].float(), L["g_idx"], L["bias"].float(), bits, mode)
        scale = float(ref.abs().max())
        e_lo, e_hi = float((y_lo.float() - ref).abs().max()) / scale, float((y_hi.float() - ref).abs().max()) / scale
        # fp32 reduce: the partials' own fp16 rounding + one final rounding; fp16 reduce: one more rounding per rank.  Stated tolerance: 2e-3 of the
        # largest output for both at T = 2 (the matmul tolerance of the GPU tests), and the one-pass form is allowed up to 2x the fp32 form's error + 1 ulp
        q.put((rank, e_lo <= 2e-3 and e_hi <= 2e-3 and e_lo <= 2 * e_hi + 1e-3 and y_lo.dtype == torch.float16 and y_hi.dtype == torch.float16, (e_lo, e_hi)))
    finally:
        dist.destroy_process_group()


def test_row_parallel_reduce_precision_threshold_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_row_threshold, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def _worker_mlp(rank, world, port, K, F, gs, M, q):
    """Megatron pairing on a gated MLP: gate/up column-parallel WITHOUT gather, down row-parallel with input_is_parallel=True:
    one all-reduce per block, no all-gather.  Rank-local matmuls are played by the oracle (fp32)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autogptq_amd.tensor_parallel import RowParallelQuantLinear, shard_packed_rows
        gate = O.random_quant_layer(K, F, 4, gs, seed=11)
        up = O.random_quant_layer(K, F, 4, gs, seed=12)
        down = O.random_quant_layer(F, K, 4, gs, seed=13, bias=True)
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(7)) - 0.5).float()
        mode = O.ZERO_WRAP

        def col(L):
            qw, qz, sc, _ = shard_packed(L["qweight"], L["qzeros"], L["scales"].float(), None, 4, rank, world)
            return ColumnParallelQuantLinear(lambda xx: O.forward(xx, qw, qz, sc, None, None, 4, mode), F, gather_output=False)

        g_mod, u_mod = col(gate), col(up)
        qw, qz, sc, (k0, k1) = shard_packed_rows(down["qweight"], down["qzeros"], down["scales"].float(), 4, gs, rank, world)
        d_mod = RowParallelQuantLinear(lambda xx: O.forward(xx, qw, qz, sc, None, None, 4, mode), (k0, k1), bias=down["bias"].float(),
                                       input_is_parallel=True)
        h = torch.nn.functional.silu(g_mod(x)) * u_mod(x)                 # [M, F / world]: stays sharded
        assert tuple(h.shape) == (M, F // world) and (k0, k1) == shard_bounds(F, rank, world)
        y = d_mod(h)

        def full(L, xx, b=None):
            return O.forward(xx, L["qweight"], L["qzeros"], L["scales"].float(), None, b, 4, mode)

        ref = full(down, torch.nn.functional.silu(full(gate, x)) * full(up, x), down["bias"].float())
        q.put((rank, bool(torch.allclose(y, ref, rtol=1e-4, atol=1e-5)), float((y - ref).abs().max())))
    finally:
        dist.destroy_process_group()


def test_column_then_row_parallel_mlp_two_ranks_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mlp, args=(r, world, port, 256, 512, 128, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
