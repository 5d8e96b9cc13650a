"""Direct peer-store all-gather (csrc/peer.hip).

CPU (-m "not gpu"): struct layout and the validation of the three entry points (nothing is launched: every case fails before
the launch).  GPU (-m gpu): T ranks simulated inside one process (each with its own buffers, flags and state on cuda:0;
scatter for all ranks, then collect for all -- the split entry points exist for exactly this and for overlapping work), a
captured graph replayed over several epochs, the bounded wait, and two processes sharing cuda:0 through CUDA-IPC mappings with
the real column-parallel shards (the only multi-process configuration a 1-GPU box offers)."""
import ctypes
import os
import queue
import socket
import time

import pytest
import torch

from autogptq_amd import _lib


def _dummy_group(world=2, rank=0, rows_max=4, N=128, null=()):
    pg = _lib.GptqPeerGroup()
    for r in range(min(max(world, 0), _lib.PEER_MAX)):
        pg.xbuf[0][r] = 0x1000
        pg.xbuf[1][r] = 0x2000
        pg.flags[r] = 0x3000
    pg.state = 0x4000
    pg.world, pg.rank, pg.rows_max, pg.N = world, rank, rows_max, N
    for name in null:
        if name == "state":
            pg.state = None
        elif name == "flags1":
            pg.flags[1] = None
        elif name == "xbuf1":
            pg.xbuf[1][0] = None
    return pg


def test_peer_group_struct_layout():
    assert _lib.PEER_MAX == 8
    assert ctypes.sizeof(_lib.GptqPeerGroup) == 2 * 8 * 8 + 8 * 8 + 8 + 4 * 4
    assert _lib.GptqPeerGroup.flags.offset == 128 and _lib.GptqPeerGroup.state.offset == 192
    assert _lib.GptqPeerGroup.world.offset == 200 and _lib.GptqPeerGroup.N.offset == 212


@pytest.mark.parametrize("kw,M,nl,status", [
    (dict(world=0), 1, 64, 2), (dict(world=9), 1, 64, 2), (dict(rank=2), 1, 64, 2), (dict(rank=-1), 1, 64, 2),
    (dict(null=("state",)), 1, 64, 1), (dict(null=("flags1",)), 1, 64, 1), (dict(null=("xbuf1",)), 1, 64, 1),
    (dict(N=96), 1, 48, 2),             # 48 columns per rank: not a multiple of 32
    (dict(N=130), 1, 65, 2),
    (dict(), 5, 64, 2),                 # M > rows_max
    (dict(), 0, 64, 2),
    (dict(), 1, 32, 2),                 # n_local != N / world
])
def test_peer_entry_points_validate_before_launching(kw, M, nl, status):
    lib = _lib.load()
    pg = _dummy_group(**kw)
    rc = lib.gptq_peer_scatter(ctypes.byref(pg), 0x5000, M, nl, _lib.GPTQ_F16, None)
    assert rc == status, lib.gptq_last_error()
    rc = lib.gptq_peer_gather(ctypes.byref(pg), 0x5000, 0x6000, M, nl, _lib.GPTQ_F16, 16, None)
    assert rc == status
    if nl == 64:
        assert lib.gptq_peer_collect(ctypes.byref(pg), 0x6000, M, _lib.GPTQ_F16, 16, None) == status


def test_peer_entry_points_null_and_unbounded_wait_are_errors():
    lib = _lib.load()
    pg = _dummy_group()
    assert lib.gptq_peer_scatter(None, 0x5000, 1, 64, 0, None) == 1
    assert lib.gptq_peer_scatter(ctypes.byref(pg), None, 1, 64, 0, None) == 1
    assert lib.gptq_peer_collect(ctypes.byref(pg), None, 1, 0, 16, None) == 1
    assert lib.gptq_peer_collect(ctypes.byref(pg), 0x6000, 1, 0, 0, None) == 2           # max_spins = 0: the wait is bounded by design
    assert b"bounded" in lib.gptq_last_error()
    assert lib.gptq_peer_gather(ctypes.byref(pg), 0x5000, 0x6000, 1, 64, 7, 16, None) == 3   # dtype enum


def test_column_parallel_rejects_unknown_exchange():
    from autogptq_amd.tensor_parallel import ColumnParallelQuantLinear
    with pytest.raises(ValueError):
        ColumnParallelQuantLinear(lambda x: x, 64, exchange="ring")


def test_ipc_mapping_helper_checks_the_torch_signature_it_patches():
    """_map overrides `storage_device` in the argument tuple of torch's CUDA-IPC reduction: by NAME, and a tuple of another arity
    (a different torch) is an error, not a silently wrong index."""
    import inspect
    from torch.multiprocessing.reductions import rebuild_cuda_tensor
    from autogptq_amd.peer_exchange import _map
    assert "storage_device" in inspect.signature(rebuild_cuda_tensor).parameters
    with pytest.raises(RuntimeError, match="signature"):
        _map((1, 2, 3), "cuda:0")


# ---- GPU ---------------------------------------------------------------------------------------------------------------
def _sim_groups(T, rows_max, N, dtype, dev):
    from autogptq_amd.peer_exchange import make_group
    x0 = [torch.full((rows_max, N), -7.0, dtype=dtype, device=dev) for _ in range(T)]
    x1 = [torch.full((rows_max, N), -9.0, dtype=dtype, device=dev) for _ in range(T)]
    fl = [torch.zeros(_lib.PEER_MAX, dtype=torch.int32, device=dev) for _ in range(T)]
    st = [torch.zeros(4, dtype=torch.int32, device=dev) for _ in range(T)]
    groups = [make_group(x0, x1, fl, st[r], r, rows_max, N) for r in range(T)]
    return groups, (x0, x1, fl, st)


@pytest.mark.gpu
@pytest.mark.parametrize("T", [1, 2, 4, 8])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M", [1, 5, 64])
def test_peer_gather_simulated_ranks(T, dtype, M):
    dev = torch.device("cuda:0")
    lib = _lib.load()
    nl = 96 if T > 1 else 128
    N = T * nl
    groups, keep = _sim_groups(T, 64, N, dtype, dev)
    st = _lib.current_stream_handle(dev)
    dt = _lib.DTYPE_ENUM[dtype]
    for epoch in range(1, 4):                                   # three calls: both parities, and a reuse of parity 1
        ys = [(torch.rand(M, nl, device=dev) * (epoch + r)).to(dtype) for r in range(T)]
        outs = [torch.zeros(M, N, dtype=dtype, device=dev) for _ in range(T)]
        for r in range(T):
            _lib.check(lib.gptq_peer_scatter(ctypes.byref(groups[r]), ys[r].data_ptr(), M, nl, dt, st))
        for r in range(T):
            _lib.check(lib.gptq_peer_collect(ctypes.byref(groups[r]), outs[r].data_ptr(), M, dt, 1 << 16, st))
        torch.cuda.synchronize()
        want = torch.cat(ys, dim=1)
        for r in range(T):
            assert torch.equal(outs[r], want), (epoch, r)
            assert keep[3][r].tolist() == [epoch, 0, 0, 0]      # epoch advanced, tickets back to zero, no timeout
            assert keep[2][r][:T].tolist() == [epoch] * T


@pytest.mark.gpu
def test_peer_gather_replays_inside_a_graph():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    T, M, nl = 4, 3, 256
    N = T * nl
    groups, keep = _sim_groups(T, 8, N, torch.float16, dev)
    ys = [torch.zeros(M, nl, dtype=torch.float16, device=dev) for _ in range(T)]
    outs = [torch.zeros(M, N, dtype=torch.float16, device=dev) for _ in range(T)]
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            st = _lib.current_stream_handle(dev)
            for r in range(T):
                _lib.check(lib.gptq_peer_scatter(ctypes.byref(groups[r]), ys[r].data_ptr(), M, nl, _lib.GPTQ_F16, st))
            for r in range(T):
                _lib.check(lib.gptq_peer_collect(ctypes.byref(groups[r]), outs[r].data_ptr(), M, _lib.GPTQ_F16, 1 << 16, st))
        for it in range(5):
            for r in range(T):
                ys[r].copy_(torch.full((M, nl), float(10 * it + r), device=dev))
            g.replay()
            s.synchronize()
            want = torch.cat(ys, dim=1)
            for r in range(T):
                assert torch.equal(outs[r], want), (it, r)
    assert keep[3][0].tolist() == [5, 0, 0, 0]


@pytest.mark.gpu
def test_peer_collect_gives_up_instead_of_hanging():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    T, M, nl = 2, 1, 128
    groups, keep = _sim_groups(T, 4, T * nl, torch.float16, dev)
    y = torch.ones(M, nl, dtype=torch.float16, device=dev)
    out = torch.zeros(M, T * nl, dtype=torch.float16, device=dev)
    st = _lib.current_stream_handle(dev)
    _lib.check(lib.gptq_peer_scatter(ctypes.byref(groups[0]), y.data_ptr(), M, nl, _lib.GPTQ_F16, st))
    _lib.check(lib.gptq_peer_collect(ctypes.byref(groups[0]), out.data_ptr(), M, _lib.GPTQ_F16, 64, st))    # rank 1 never scatters
    torch.cuda.synchronize()
    assert keep[3][0].tolist() == [1, 0, 0, 1]                  # timeout raised, epoch still advanced, kernel returned
    assert torch.equal(out[:, :nl], y)                          # own slice is there, the peer's is whatever the buffer held


def _shards(K, N, T, bits, gs, dtype, dev, seed):
    """The T column shards of one random layer (each an mi355x QuantLinear with its decode copy) + the oracle's full dequantised weight."""
    from autogptq_amd import QuantLinear
    from autogptq_amd.tensor_parallel import ColumnParallelQuantLinear
    from oracle import gptq_oracle as O
    L = O.random_quant_layer(K, N, bits, gs, dtype=dtype, seed=seed, bias=True)
    full = QuantLinear(bits, gs, K, N, True, weight_dtype=dtype)
    full.qweight, full.qzeros, full.scales, full.g_idx, full.bias = L["qweight"], L["qzeros"], L["scales"], L["g_idx"], L["bias"]
    shards = [ColumnParallelQuantLinear.from_full(full, r, T, device=dev, gather_output=False).local for r in range(T)]
    for q in shards:
        q.post_init()
    mode = O.ZERO_NOWRAP if bits == 3 else O.ZERO_WRAP
    W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], bits, mode).to(dev)
    return shards, W, L["bias"].to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize("T,K,N,bits,dtype", [(2, 1024, 2048, 4, torch.float16), (4, 2048, 1024, 4, torch.bfloat16), (8, 8192, 8192, 4, torch.float16),
                                             (8, 4096, 2048, 8, torch.float16), (4, 2048, 2048, 3, torch.float16)])
def test_forward_scatter_is_the_kernel_epilogue(T, K, N, bits, dtype):
    """gptq_forward_scatter on T simulated ranks (round 4): every rank's decode-copy kernel stores its slice into every rank's exchange buffer from its own
    epilogue, then ONE collect per rank.  The gathered rows equal the concatenation of the shards' plain forwards BIT FOR BIT (same kernel, same plan) and
    every output agrees with x (fp64) @ W_oracle (fp64); three epochs (both parities + a reuse); tickets back at zero; (8, 8192, 8192): the 70B attention
    shard, whose 64 strips run as K slices combined inside the launch before the owner slice scatters."""
    from autogptq_amd.qlinear_mi355x import reserve_workspace
    dev = torch.device("cuda:0")
    lib = _lib.load()
    shards, W, bias = _shards(K, N, T, bits, 128, dtype, dev, 5 + T)
    nl = N // T
    assert all(q._qweight_tiled is not None for q in shards)
    for M in (1, 3, 4):
        groups, keep = _sim_groups(T, 4, N, dtype, dev)
        st = _lib.current_stream_handle(dev)
        need = max(int(lib.gptq_workspace_bytes(ctypes.byref(q._layer), M)) for q in shards)
        ws = reserve_workspace(dev, need) if need else None
        for epoch in range(1, 4):
            x = ((torch.rand(M, K, generator=torch.Generator().manual_seed(100 * epoch + M)) - 0.5)).to(dtype).to(dev)
            outs = [torch.zeros(M, N, dtype=dtype, device=dev) for _ in range(T)]
            for r in range(T):
                _lib.check(lib.gptq_forward_scatter(ctypes.byref(shards[r]._layer), x.data_ptr(), M, ctypes.byref(groups[r]),
                                                    ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, st))
            for r in range(T):                                  # the simulated ranks share ONE stream: their flags are raised before anyone's collect waits
                _lib.check(lib.gptq_peer_publish(ctypes.byref(groups[r]), st))
            for r in range(T):
                _lib.check(lib.gptq_peer_collect(ctypes.byref(groups[r]), outs[r].data_ptr(), M, _lib.DTYPE_ENUM[dtype], 1 << 16, st))
            torch.cuda.synchronize()
            with torch.no_grad():
                want = torch.cat([q(x) for q in shards], dim=1)
            ref = x.double() @ W.double() + bias.double()
            rtol, atol = (1e-3, 1e-3) if dtype == torch.float16 else (8e-3, 8e-3)
            assert not bool(((want.double() - ref).abs() > atol * float(ref.abs().max()) + rtol * ref.abs()).any())
            for r in range(T):
                assert torch.equal(outs[r], want), (M, epoch, r)
                assert keep[3][r].tolist() == [epoch, 0, 0, 0]
                assert keep[2][r][:T].tolist() == [epoch] * T
    if (T, K) == (8, 8192):
        # (until late round 6 these 64-strip shards ran as K slices -- the combination fused scatter + in-launch slice exchange; the planner now takes no slices
        # below 5120 k per slice: tools/tp_shard_sweep.py.  The K-sliced peer form stays covered by the forced plans of tests/test_gpu_tiled.py)
        assert _lib.describe_plan(shards[0]._layer, 1)["ksplit"] == 1


@pytest.mark.gpu
def test_forward_gather_replays_inside_a_graph_and_refuses_what_it_cannot_fuse():
    from autogptq_amd.qlinear_mi355x import reserve_workspace
    dev = torch.device("cuda:0")
    lib = _lib.load()
    T, K, N, M = 4, 4096, 4096, 2
    shards, W, bias = _shards(K, N, T, 4, 128, torch.float16, dev, 77)
    groups, keep = _sim_groups(T, 4, N, torch.float16, dev)
    x = torch.zeros(M, K, dtype=torch.float16, device=dev)
    outs = [torch.zeros(M, N, dtype=torch.float16, device=dev) for _ in range(T)]
    need = max(int(lib.gptq_workspace_bytes(ctypes.byref(q._layer), M)) for q in shards)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        ws = reserve_workspace(dev, max(need, 1 << 17))
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            st = _lib.current_stream_handle(dev)
            for r in range(T):
                _lib.check(lib.gptq_forward_scatter(ctypes.byref(shards[r]._layer), x.data_ptr(), M, ctypes.byref(groups[r]), ws.data_ptr(), ws.numel(), st))
            for r in range(T):
                _lib.check(lib.gptq_peer_publish(ctypes.byref(groups[r]), st))
            for r in range(T):
                _lib.check(lib.gptq_peer_collect(ctypes.byref(groups[r]), outs[r].data_ptr(), M, _lib.GPTQ_F16, 1 << 16, st))
        for it in range(5):
            x.copy_(((torch.rand(M, K, generator=torch.Generator().manual_seed(it)) - 0.5)).half())
            g.replay()
            s.synchronize()
            with torch.no_grad():
                want = torch.cat([q(x) for q in shards], dim=1)
            for r in range(T):
                assert torch.equal(outs[r], want), (it, r)
    assert keep[3][0].tolist() == [5, 0, 0, 0]
    # what the fused form does not cover is an error the caller falls back from (ColumnParallelQuantLinear: local forward + scatter + collect)
    st = _lib.current_stream_handle(dev)
    x8 = torch.zeros(8, K, dtype=torch.float16, device=dev)
    g8, _ = _sim_groups(T, 8, N, torch.float16, dev)
    assert lib.gptq_forward_scatter(ctypes.byref(shards[0]._layer), x8.data_ptr(), 8, ctypes.byref(g8[0]), ws.data_ptr(), ws.numel(), st) == 3
    assert b"gptq_peer_scatter" in lib.gptq_last_error()
    gbad, _ = _sim_groups(2, 4, N, torch.float16, dev)                # N / world != the shard's width
    assert lib.gptq_forward_scatter(ctypes.byref(shards[0]._layer), x.data_ptr(), 1, ctypes.byref(gbad[0]), ws.data_ptr(), ws.numel(), st) == 2
    assert lib.gptq_forward_gather(ctypes.byref(shards[0]._layer), x.data_ptr(), outs[0].data_ptr(), 1, ctypes.byref(groups[0]), 0, ws.data_ptr(), ws.numel(), st) == 2
    assert lib.gptq_forward_scatter(None, x.data_ptr(), 1, ctypes.byref(groups[0]), None, 0, st) == 1


@pytest.mark.gpu
def test_fine_grained_buffer_is_plain_device_memory_for_torch():
    """hipExtMallocWithFlags(finegrained) memory wrapped as a tensor: zeroed, writable by torch kernels, visible through a second alias of
    the same pointer, and PeerExchange uses it by default (one rank: nothing to map)."""
    from autogptq_amd.peer_exchange import FineGrainedBuffer, PeerExchange
    dev = torch.device("cuda:0")
    fb = FineGrainedBuffer(4096 * 2, dev)
    t = fb.tensor((4, 1024), torch.float16)
    assert t.device == dev and int(t.count_nonzero()) == 0
    t.copy_(torch.arange(4096, dtype=torch.float32, device=dev).reshape(4, 1024).half())
    u = fb.tensor((4096,), torch.float16)
    torch.cuda.synchronize()
    assert float(u[1025]) == 1025.0 and u.data_ptr() == fb.ptr
    b = fb.tensor((2048,), torch.bfloat16)
    assert b.dtype == torch.bfloat16 and b.data_ptr() == fb.ptr
    assert len(fb.ipc_handle()) == 64
    px = PeerExchange(4, 256, torch.float16, dev)
    assert px.fine_grained and not px.multi_device
    y = torch.ones(2, 256, dtype=torch.float16, device=dev)
    out = px.gather(y)
    px.check_timeout()
    assert torch.equal(out, y)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ipc_worker(rank, world, port, M, q, act=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autogptq_amd import QuantLinear
        from autogptq_amd.tensor_parallel import ColumnParallelQuantLinear
        from oracle import gptq_oracle as O
        dev = "cuda:0"
        K, N = 1024, 2048
        L = O.random_quant_layer(K, N, 4, 128, act_order=act, seed=11)                   # identical on every rank
        m = QuantLinear(4, 128, K, N, False)
        m.qweight, m.qzeros, m.scales, m.g_idx = L["qweight"], L["qzeros"], L["scales"], L["g_idx"]
        cp = ColumnParallelQuantLinear.from_full(m, rank, world, device=dev, gather_output=True, exchange="peer_store", max_rows=M)
        errs = []
        with torch.no_grad():
            for it in range(3):
                x = (torch.rand(M, K, generator=torch.Generator().manual_seed(20 + it)) - 0.5).half()
                y = cp(x.to(dev))
                cp._px.check_timeout()
                y64 = O.forward_f64(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"] if act else None, None, 4, O.ZERO_NOWRAP if act else O.ZERO_WRAP)
                errs.append(float((y.double().cpu() - y64).abs().max() / y64.abs().max()))
        dist.barrier()                                              # nobody unmaps while a peer may still store
        # decode rows of a plain shard: the scatter is the shard kernel's epilogue (gptq_forward_scatter); act-order shards (their decode kernel gathers x through
        # perm and has no scatter epilogue) take local forward + scatter + collect
        fused_as_expected = cp.fused_calls == (3 if (M <= 4 and not act) else 0)
        q.put((rank, tuple(y.shape) == (M, N) and max(errs) < 3e-3 and cp._px.fine_grained and fused_as_expected,
               errs + [("fine_grained", cp._px.fine_grained), ("fused_calls", cp.fused_calls)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("M,act", [(1, False), (48, False), (1, True), (4, True)], ids=["M1", "M48", "M1-act", "M4-act"])
def test_two_processes_one_gpu_peer_store_column_parallel(M, act):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ipc_worker, args=(r, world, port, M, q, act)) for r in range(world)]
    for p in procs:
        p.start()
    res, deadline = [], time.time() + 150
    while len(res) < world:                        # a crashed worker fails the test at once instead of sitting out a queue timeout
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > deadline:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                pytest.fail(f"peer-store workers: exit codes {[p.exitcode for p in procs]}, results so far {res}")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def _two_gpu_worker(rank, world, port, exchange, calls, q):
    """One rank per REAL device: the column shard's decode kernel with the fused scatter storing into the peer's exchange buffer across xGMI (peer_store), or
    RCCL's all_gather_into_tensor (all_gather) -- every output of every call against the oracle, the sticky timeout word read at the end."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = f"cuda:{rank}"
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    try:
        from autogptq_amd import QuantLinear
        from autogptq_amd.tensor_parallel import ColumnParallelQuantLinear
        from oracle import gptq_oracle as O
        K, N = 4096, 4096
        L = O.random_quant_layer(K, N, 4, 128, act_order=False, seed=21)                 # identical on every rank
        m = QuantLinear(4, 128, K, N, False)
        m.qweight, m.qzeros, m.scales, m.g_idx = L["qweight"], L["qzeros"], L["scales"], L["g_idx"]
        W = O.dequantize(L["qweight"], L["qzeros"], L["scales"], L["g_idx"], 4, O.ZERO_WRAP).to(dev).double()
        cp = ColumnParallelQuantLinear.from_full(m, rank, world, device=dev, gather_output=True, exchange=exchange, max_rows=4)
        worst, bad_calls = 0.0, 0
        with torch.no_grad():
            for it in range(calls):
                M = 1 + it % 4
                x = (torch.rand(M, K, generator=torch.Generator().manual_seed(100 + it)) - 0.5).half().to(dev)
                y = cp(x)
                if it % 50 == 0 or it >= calls - 8:                                  # every output, on a sample of the calls and the last ones back to back
                    ref = x.double() @ W
                    err = float((y.double() - ref).abs().max() / ref.abs().max())
                    worst = max(worst, err)
                    bad_calls += int(err > 3e-3)
        if exchange == "peer_store":
            cp._px.check_timeout()                                                   # raises if any collect gave up on the peer
        dist.barrier()
        fused_ok = exchange != "peer_store" or cp.fused_calls == calls
        q.put((rank, tuple(y.shape) == (M, N) and bad_calls == 0 and fused_ok, dict(worst=worst, bad_calls=bad_calls, fused_calls=getattr(cp, "fused_calls", None))))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the first multi-GPU lease runs it by itself)")
@pytest.mark.parametrize("exchange", ["all_gather", "peer_store"])
def test_two_gpus_column_parallel_across_xgmi(exchange):
    """The first test that runs on two REAL devices: RCCL all-gather and the direct peer-store exchange with the scatter fused into the decode kernel's epilogue
    (csrc/gemv_tiled_kernel.cuh: sc0 sc1 write-through stores into the PEER's buffer, the flag raised by the collect launch) -- 1,000 back-to-back calls,
    outputs against the oracle, the sticky error word checked.  Every earlier TP test ran with both ranks on one device, where "peer" memory is local."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_gpu_worker, args=(r, world, port, exchange, 1000, q)) for r in range(world)]
    for p in procs:
        p.start()
    res, deadline = [], time.time() + 300
    while len(res) < world:
        try:
            res.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() > deadline:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                pytest.fail(f"two-GPU workers: exit codes {[p.exitcode for p in procs]}, results so far {res}")
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res
