"""GPU (-m gpu): the tensor-parallel wrappers together with the real HIP shards.  Two ranks (gloo control plane, exchanges
staged through host memory) share cuda:0 -- RCCL refuses two ranks on one GPU, and an 8-GPU node is the driver's to launch --
so this proves wrapper + kernels + exchange logic end to end; the RCCL transport itself is exercised by bench.py --gpus N.

Layout under test = SURVEY 8(e) + f4: column-parallel gate / up WITHOUT gather feeding a row-parallel down projection
(one all-reduce per MLP block), and a gathered column-parallel attention projection."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, M, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from autogptq_amd import QuantLinear
        from autogptq_amd.tensor_parallel import ColumnParallelQuantLinear, RowParallelQuantLinear
        from oracle import gptq_oracle as O
        dev = "cuda:0"
        K, I = 1024, 2048                                       # hidden, intermediate (per rank: 1024 columns / rows)

        def full(Kf, Nf, seed, act=False):
            L = O.random_quant_layer(Kf, Nf, 4, 128, act_order=act, seed=seed)       # identical on every rank
            m = QuantLinear(4, 128, Kf, Nf, False)
            m.qweight, m.qzeros, m.scales, m.g_idx = L["qweight"], L["qzeros"], L["scales"], L["g_idx"]
            return L, m

        Lg, gate = full(K, I, 1)
        Lu, up = full(K, I, 2)
        Ld, down = full(I, K, 3)
        La, attn = full(K, K, 4, act=True)
        x = (torch.rand(M, K, generator=torch.Generator().manual_seed(9)) - 0.5).half()
        cg = ColumnParallelQuantLinear.from_full(gate, rank, world, device=dev, gather_output=False)
        cu = ColumnParallelQuantLinear.from_full(up, rank, world, device=dev, gather_output=False)
        rd = RowParallelQuantLinear.from_full(down, rank, world, device=dev, input_is_parallel=True)
        ca = ColumnParallelQuantLinear.from_full(attn, rank, world, device=dev, gather_output=True)
        with torch.no_grad():
            xd = x.to(dev)
            h = torch.nn.functional.silu(cg(xd).float()).to(torch.float16) * cu(xd)      # [M, I / world], never gathered
            y = rd(h)                                                                  # one all-reduce
            ya = ca(xd)                                                                # one all-gather
        # reference: fp64 oracle of the unsharded layers
        g64 = O.forward_f64(x, Lg["qweight"], Lg["qzeros"], Lg["scales"], None, None, 4, O.ZERO_WRAP)
        u64 = O.forward_f64(x, Lu["qweight"], Lu["qzeros"], Lu["scales"], None, None, 4, O.ZERO_WRAP)
        h64 = (torch.nn.functional.silu(g64) * u64)
        y64 = O.forward_f64(h64, Ld["qweight"], Ld["qzeros"], Ld["scales"], None, None, 4, O.ZERO_WRAP)
        a64 = O.forward_f64(x, La["qweight"], La["qzeros"], La["scales"], La["g_idx"], None, 4, O.ZERO_NOWRAP)
        e_mlp = float((y.double().cpu() - y64).abs().max() / y64.abs().max())
        e_att = float((ya.double().cpu() - a64).abs().max() / a64.abs().max())
        ok = tuple(y.shape) == (M, K) and tuple(ya.shape) == (M, K) and e_mlp < 6e-3 and e_att < 3e-3
        q.put((rank, bool(ok), e_mlp, e_att))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("M", [1, 48])
def test_two_ranks_one_gpu_column_row_mlp_and_gathered_attention(M):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _ in res), res


def test_bench_two_ranks_headline_is_the_tensor_parallel_stack():
    """`bench.py --gpus 2` (round 6): the line's `value` is the TENSOR-PARALLEL stack of BASELINE config 4 (Llama-2-70B shapes, every layer split over
    out_features + one all-gather per layer: north_star's partitioning), the data-parallel replicas are a secondary object, and the line says what the process
    group saw (world size, device per rank, exchange, backend).  Run exactly as the driver launches it -- torch.distributed.run, one process per rank -- with both
    ranks on cuda:0 and a gloo control plane (RCCL refuses two ranks on one GPU); the JSON shape is what is asserted, no multi-GPU number is claimed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(root, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--blocks", "1", "--tp-blocks", "1", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-tp-layers", "--no-fused", "--no-extras"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["unit"] == "GB/s" and d["value"] > 0 and d["higher_is_better"] is True, d
    cfg = d["config"]
    assert cfg["parallelism"] == "tp2" and cfg["nccl_world"] == 2 and cfg["backend"] == "gloo" and cfg["exchange"] == "all_gather", cfg
    assert len(cfg["rank_devices"]) == 2 and {e["rank"] for e in cfg["rank_devices"]} == {0, 1} and all(e["device"] == "cuda:0" for e in cfg["rank_devices"]), cfg
    assert "Llama-2-70B" in cfg["workload"] and cfg["layers"] == 7 and "TP=2" in d["metric"], (cfg, d["metric"])
    st = d["tp"]["stack"]
    assert abs(d["value"] - st["GB_per_s"]) < 1e-6 and abs(d["ms_per_step"] - st["ms_per_step"]) < 1e-9
    assert st["ms_per_step"] >= st["ms_per_step_local_kernels_only"] * 0.5 and st["algorithmic_bytes_per_step_per_rank"] * 2 == d["algorithmic_bytes_per_step"]
    dp = d["dp_replicas"]
    assert dp["unit"] == "GB/s" and dp["value"] > 0 and "Llama-7B" in dp["workload"]
    roof = d["roofline"]
    for k in ("tp2_stack_ms_per_step", "tp2_stack_exchange_ms_per_step", "tp2_stack_GB_per_s", "tp2_stack_GB_per_s_per_gpu", "dp_value_GB_per_s"):
        assert k in roof, (k, sorted(roof)[:40])
