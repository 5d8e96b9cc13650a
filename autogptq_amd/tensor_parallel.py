"""out_features (column) tensor parallelism for QuantLinear: one process per GPU, one exchange.

The reference has no multi-GPU execution of a layer at all (accelerate layer placement only,
auto_gptq/modeling/_utils.py:341-377; ``test_multigpu`` is an empty TODO, tests/test_q4.py:1224-1226).
What makes the split legal is the layout itself: output column n depends only on column n of
``qweight``/``scales`` and on field n of ``qzeros[:, n // P]`` -- the reference relies on the same
fact when it concatenates packed q/k/v along dim=1 (auto_gptq/nn_modules/fused_llama_attn.py:171-186).

Rank r of T owns columns [r*N/T, (r+1)*N/T): ``qweight[:, n0:n1]``, ``qzeros[:, n0*bits/32 : n1*bits/32]``,
``scales[:, n0:n1]``, ``bias[n0:n1]``; ``g_idx`` and x are replicated.  (N/T) must be a multiple of
32 (the 3-bit packing unit).  forward = local GEMV/GEMM + ONE all-gather of the [M, N/T] outputs
(RCCL over xGMI when the process group backend is "nccl").

``exchange="peer_store"`` (experimental, off by default) replaces the collective by the direct peer-store
exchange of ``csrc/peer.hip``: every rank stores its slice into every rank's exchange buffer (one xGMI link per peer), raises a
flag, and copies its own gathered rows out -- two launches, graph-capturable, no rank-major -> column-major copy for M > 1.
For decode rows (M <= 4) on a shard that carries its decode copy the scatter is the EPILOGUE of the shard's kernel (gptq_forward_scatter): a
tensor-parallel layer is then the local kernel + one collect launch.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


def _host_staged(group) -> bool:
    """gloo has no device collectives for every op used here: with a gloo group the exchange is staged through host memory
    (CPU control plane, e.g. the 2-ranks-on-one-GPU test); RCCL ("nccl") groups exchange device buffers directly."""
    return dist.get_backend(group) == "gloo"


def shard_bounds(N: int, rank: int, world: int):
    if N % world != 0 or (N // world) % 32 != 0:
        raise ValueError(f"out_features={N} cannot be split over {world} ranks in multiples of 32 columns")
    n = N // world
    return rank * n, (rank + 1) * n


def shard_packed(qweight, qzeros, scales, bias, bits: int, rank: int, world: int):
    """Column slice of the four checkpoint tensors for ``rank`` (contiguous copies)."""
    N = qweight.shape[1]
    n0, n1 = shard_bounds(N, rank, world)
    z0, z1 = n0 * bits // 32, n1 * bits // 32
    return (qweight[:, n0:n1].contiguous(), qzeros[:, z0:z1].contiguous(), scales[:, n0:n1].contiguous(),
            None if bias is None else bias[n0:n1].contiguous())


class ColumnParallelQuantLinear(nn.Module):
    """Wraps the rank-local shard (any module mapping [M,K] -> [M,N/T]) and gathers the output."""

    def __init__(self, local: Callable[[torch.Tensor], torch.Tensor], outfeatures: int,
                 group: Optional[dist.ProcessGroup] = None, gather_output: bool = True,
                 exchange: str = "all_gather", max_rows: int = 1, check_timeout_every: int = 256):
        super().__init__()
        if exchange not in ("all_gather", "peer_store"):
            raise ValueError(f"exchange must be 'all_gather' or 'peer_store', got {exchange!r}")
        self.local = local
        self.outfeatures = outfeatures
        self.group = group
        self.gather_output = gather_output
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.exchange = exchange
        self.max_rows = max_rows
        self.check_timeout_every = check_timeout_every
        self._calls = 0
        self.fused_calls = 0                # forwards that ran as local kernel (scatter in its epilogue) + collect
        self._px = None                     # PeerExchange, built on the first forward (needs the output dtype / device)

    @classmethod
    def from_full(cls, full, rank: int, world: int, group=None, device=None, gather_output=True,
                  exchange: str = "all_gather", max_rows: int = 1, check_timeout_every: int = 256):
        """Build the rank's shard from a full (unsharded) mi355x QuantLinear."""
        from .qlinear_mi355x import QuantLinear

        if getattr(full, "epilogue", "none") != "none":
            # a plain column split of [gate | up] gives rank 0 only gate columns: the SiLU*mul pairing cannot be applied locally
            raise ValueError("column-parallel split of a layer with a fused epilogue is not supported; shard gate and up separately")
        qw, qz, sc, b = shard_packed(full.qweight, full.qzeros, full.scales, full.bias, full.bits, rank, world)
        local = QuantLinear(full.bits, full.group_size, full.infeatures, qw.shape[1], b is not None,
                            weight_dtype=full.scales.dtype, zero_mode=full.zero_mode)
        local.qweight, local.qzeros, local.scales, local.g_idx = qw, qz, sc, full.g_idx.clone()
        if b is not None:
            local.bias = b
        if device is not None:
            local = local.to(device)
        return cls(local, full.outfeatures, group=group, gather_output=gather_output, exchange=exchange, max_rows=max_rows,
                   check_timeout_every=check_timeout_every)

    def _fused_peer_forward(self, x: torch.Tensor):
        """Decode rows on a shard that carries its decode copy: the scatter is the kernel's epilogue (gptq_forward_scatter) -- local kernel + one collect.
        None when this call does not qualify (then: local forward + scatter + collect)."""
        q = self.local
        if self.exchange != "peer_store" or not self.gather_output or self.world == 1 or not x.is_cuda or not hasattr(q, "_qweight_tiled"):
            return None
        K = q.infeatures
        M = x.numel() // K if K else 0
        if M < 1 or M > 4 or x.dtype != q.scales.dtype:
            return None
        if q._layer is None:
            q.post_init()
        nl = q.outfeatures
        if self._px is None or self._px.rows_max < M:
            from .peer_exchange import PeerExchange
            self._px = PeerExchange(max(M, self.max_rows), self.world * nl, q.scales.dtype, x.device, group=self.group)
        if not self._px.fused_ok(q, M):
            return None
        x2 = x.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        out = self._px.forward_gather(q, x2).reshape(x.shape[:-1] + (self.world * nl,))
        self.fused_calls += 1
        self._calls += 1
        if self.check_timeout_every and self._calls % self.check_timeout_every == 0:
            self._px.check_timeout()
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        fused = self._fused_peer_forward(x)
        if fused is not None:
            return fused
        y = self.local(x)                                   # [..., N/T]
        if not self.gather_output or self.world == 1:
            return y
        lead = y.shape[:-1]
        y2 = y.reshape(-1, y.shape[-1]).contiguous()         # [M, N/T]
        M, nl = y2.shape
        if self.exchange == "peer_store":
            if not y2.is_cuda:
                raise RuntimeError("exchange='peer_store' moves device buffers; the shard output is on the CPU")
            if self._px is None or self._px.rows_max < M:
                # collective: every rank reaches this with the same M (the buffers are symmetric).  PeerExchange allocates fine-grained
                # buffers itself and refuses a multi-GPU group on coarse-grained memory.
                from .peer_exchange import PeerExchange
                self._px = PeerExchange(max(M, self.max_rows), self.world * nl, y2.dtype, y2.device, group=self.group)
            out = self._px.gather(y2).reshape(lead + (self.world * nl,))
            # a collect that gave up on a peer leaves a partially stale row and raises state[3]: look at it every `check_timeout_every`
            # calls (a device -> host read: a sync point, hence not on every token by default; 1 = every call, 0 = never)
            self._calls += 1
            if self.check_timeout_every and self._calls % self.check_timeout_every == 0:
                self._px.check_timeout()
            return out
        if y2.is_cuda and _host_staged(self.group):
            hb = torch.empty((self.world * M, nl), dtype=y2.dtype)
            dist.all_gather_into_tensor(hb, y2.cpu(), group=self.group)
            buf = hb.to(y2.device)
        else:
            buf = torch.empty((self.world * M, nl), dtype=y2.dtype, device=y2.device)    # rank-major concatenation
            dist.all_gather_into_tensor(buf, y2, group=self.group)
        buf = buf.view(self.world, M, nl)
        if M == 1:
            out = buf.reshape(1, self.world * nl)            # rank-major == column-major for one row
        else:
            out = buf.permute(1, 0, 2).reshape(M, self.world * nl)
        return out.reshape(lead + (self.world * nl,))


def shard_packed_rows(qweight, qzeros, scales, bits: int, group_size: int, rank: int, world: int):
    """Row (in_features) slice for ``rank``: K/T input features = whole groups and whole packing units."""
    K = qweight.shape[0] * 32 // bits
    if K % world != 0 or (K // world) % group_size != 0 or (K // world) % 32 != 0:
        raise ValueError(f"in_features={K} cannot be split over {world} ranks in whole groups of {group_size} (and 32-value units)")
    k0, k1 = rank * (K // world), (rank + 1) * (K // world)
    r0, r1 = k0 * bits // 32, k1 * bits // 32
    g0, g1 = k0 // group_size, k1 // group_size
    return qweight[r0:r1].contiguous(), qzeros[g0:g1].contiguous(), scales[g0:g1].contiguous(), (k0, k1)


class RowParallelQuantLinear(nn.Module):
    """The pair of ColumnParallelQuantLinear (Megatron style): the layer is split along in_features, every rank multiplies
    its slice of x (the un-gathered output of a preceding column-parallel layer) and ONE all-reduce sums the partial
    outputs; bias is added after the reduction.  Sequential groups only (an act-order g_idx mixes groups across the whole K)."""

    def __init__(self, local: Callable[[torch.Tensor], torch.Tensor], k_range, bias: Optional[torch.Tensor] = None,
                 group: Optional[dist.ProcessGroup] = None, input_is_parallel: bool = True, fp32_reduce_max_elems: Optional[int] = 64 * 16384):
        """fp32_reduce_max_elems: outputs of up to this many elements are summed over the ranks in fp32 and rounded ONCE (decode rows, batched decode);
        larger (prefill-sized) ones are all-reduced in the layer dtype in one pass -- Megatron's row-parallel choice: half the bytes on the links, but
        each of the T partial sums is rounded before the sum (for K / T terms of |x w| <= 1 that is T half-ulp errors: ~2e-3 relative at T = 8 in fp16)
        and a partial sum beyond 65504 overflows fp16 where the fp32 total would not.  None = always fp32.  bf16 layers cannot overflow this way."""
        super().__init__()
        self.local = local
        self.k0, self.k1 = k_range
        self.bias = bias
        self.group = group
        self.input_is_parallel = input_is_parallel
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.fp32_reduce_max_elems = fp32_reduce_max_elems   # default: up to 64 rows x 16 K columns take the fp32 reduction

    @classmethod
    def from_full(cls, full, rank: int, world: int, group=None, device=None, input_is_parallel=True, fp32_reduce_max_elems: Optional[int] = 64 * 16384):
        from .qlinear_mi355x import QuantLinear, _is_sequential_g_idx

        if not _is_sequential_g_idx(full.g_idx, full.group_size):
            raise ValueError("row-parallel split needs sequential groups (no act-order)")
        if getattr(full, "epilogue", "none") != "none":
            raise ValueError("row-parallel split of a layer with a fused epilogue is not supported")
        qw, qz, sc, (k0, k1) = shard_packed_rows(full.qweight, full.qzeros, full.scales, full.bits, full.group_size, rank, world)
        local = QuantLinear(full.bits, full.group_size, k1 - k0, full.outfeatures, False, weight_dtype=full.scales.dtype,
                            zero_mode=full.zero_mode)
        local.qweight, local.qzeros, local.scales = qw, qz, sc
        bias = full.bias
        if device is not None:
            local = local.to(device)
            bias = None if bias is None else bias.to(device)
        return cls(local, (k0, k1), bias=bias, group=group, input_is_parallel=input_is_parallel, fp32_reduce_max_elems=fp32_reduce_max_elems)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not self.input_is_parallel:
            x = x[..., self.k0:self.k1].contiguous()
        y = self.local(x)
        if (self.world > 1 and self.fp32_reduce_max_elems is not None and y.numel() > self.fp32_reduce_max_elems
                and not (y.is_cuda and _host_staged(self.group))):
            # prefill-sized outputs: reduce in the layer dtype (what Megatron's row-parallel linear does) -- ONE pass over [M, N] instead
            # of cast-up, fp32 all-reduce (twice the bytes on xGMI) and cast-down; the T partial sums are each rounded once more
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
        elif self.world > 1:
            y = y.float()                                   # decode-sized outputs: partial sums are reduced in fp32, rounded once
            if y.is_cuda and _host_staged(self.group):
                h = y.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                y = h.to(y.device)
            else:
                dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)
            y = y.to(x.dtype)
        if self.bias is not None:
            y = y + self.bias
        return y


__all__ = ["ColumnParallelQuantLinear", "RowParallelQuantLinear", "shard_packed", "shard_packed_rows", "shard_bounds"]
