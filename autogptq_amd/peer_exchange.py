"""Direct peer-store all-gather for column-parallel layers; experimental, off by default.

``PeerExchange`` owns what ``gptq_peer_group_t`` (include/gptq_mi355x.h) describes: per rank two exchange buffers
``[rows_max, N]``, the arrival flags and the 4-word state, plus the peers' buffers and flags mapped into this process.
PyTorch does the plumbing only -- device memory, and its CUDA-IPC tensor sharing (``torch.multiprocessing.reductions``)
to map a peer's allocation -- the exchange itself is the two launches of ``csrc/peer.hip`` through the C ABI.

The reference has nothing to mirror here: it runs a layer on one GPU (tests/test_q4.py:1224-1226 ``test_multigpu`` is a
TODO).  What was exercised: ranks sharing ONE device (in-process groups and 2 processes through IPC, tests/).  Across
GPUs the flags and buffers have to be fine-grained allocations, which the caching allocator does not hand out: pass them
in through ``buffers=`` in that case.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib

DEFAULT_MAX_SPINS = 1 << 22          # bounded wait: ~seconds of polling, then state[3] is raised instead of hanging the queue


class PeerTimeout(RuntimeError):
    """A collect gave up waiting for a peer's slice (``state[3]`` raised by the kernel)."""


def _share(t: torch.Tensor):
    from torch.multiprocessing.reductions import reduce_tensor
    fn, args = reduce_tensor(t)
    return args


def _map(args, device):
    """Map a peer's allocation (the argument tuple of torch's CUDA-IPC reduction) into this process, on OUR device: the kernels of
    this rank store into it, so the mapping has to live in this rank's address space whichever GPU owns the memory."""
    import inspect
    from torch.multiprocessing.reductions import rebuild_cuda_tensor
    names = list(inspect.signature(rebuild_cuda_tensor).parameters)
    if len(args) != len(names) or "storage_device" not in names:
        raise RuntimeError("torch.multiprocessing.reductions.rebuild_cuda_tensor changed its signature; "
                           "pass pre-mapped buffers to PeerExchange(buffers=...) instead")
    args = list(args)
    idx = torch.device(device).index
    args[names.index("storage_device")] = idx if idx is not None else torch.cuda.current_device()
    return rebuild_cuda_tensor(*args)


class PeerExchange:
    """Symmetric exchange buffers of one process group for gathers of up to ``rows_max`` rows of ``N`` columns."""

    def __init__(self, rows_max: int, N: int, dtype: torch.dtype, device, group: Optional[dist.ProcessGroup] = None,
                 buffers: Optional[dict] = None, max_spins: int = DEFAULT_MAX_SPINS):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.world > _lib.PEER_MAX:
            raise ValueError(f"peer-store exchange supports up to {_lib.PEER_MAX} ranks, got {self.world}")
        self.rows_max, self.N, self.dtype, self.device = rows_max, N, dtype, torch.device(device)
        self.max_spins = max_spins
        if buffers is None:
            buffers = dict(xbuf0=torch.empty((rows_max, N), dtype=dtype, device=device),
                           xbuf1=torch.empty((rows_max, N), dtype=dtype, device=device),
                           flags=torch.zeros(_lib.PEER_MAX, dtype=torch.int32, device=device))
        self.state = torch.zeros(4, dtype=torch.int32, device=device)
        self._own = buffers
        torch.cuda.synchronize(self.device)
        if self.world == 1:
            mapped = [buffers]
        else:
            mine = {k: _share(v) for k, v in buffers.items()}
            everyone: List = [None] * self.world
            dist.all_gather_object(everyone, mine, group=group)
            mapped = [buffers if r == self.rank else {k: _map(a, self.device) for k, a in everyone[r].items()}
                      for r in range(self.world)]
        self._mapped = mapped                      # keeps the peers' mappings alive
        self.pg = make_group([m["xbuf0"] for m in mapped], [m["xbuf1"] for m in mapped], [m["flags"] for m in mapped],
                             self.state, self.rank, rows_max, N)
        if self.world > 1:
            dist.barrier(group=group)              # nobody scatters into a buffer that is not mapped everywhere yet

    def gather(self, y_local: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """y_local [M, N / world] -> [M, N] (every rank's slice at its column offset)."""
        M, nl = y_local.shape
        if out is None:
            out = torch.empty((M, self.N), dtype=self.dtype, device=self.device)
        st = _lib.current_stream_handle(self.device)
        _lib.check(_lib.load().gptq_peer_gather(ctypes.byref(self.pg), y_local.data_ptr(), out.data_ptr(), M, nl,
                                                _lib.DTYPE_ENUM[self.dtype], self.max_spins, st))
        return out

    def check_timeout(self) -> None:
        """Synchronises; raises if any collect so far gave up on a peer."""
        if int(self.state[3].item()) != 0:
            raise PeerTimeout("a peer-store collect timed out waiting for a peer's slice; the gathered output is incomplete")


def make_group(xbuf0: Sequence[torch.Tensor], xbuf1: Sequence[torch.Tensor], flags: Sequence[torch.Tensor],
               state: torch.Tensor, rank: int, rows_max: int, N: int) -> "_lib.GptqPeerGroup":
    """gptq_peer_group_t from tensors listed in rank order (the caller keeps them alive)."""
    pg = _lib.GptqPeerGroup()
    for r in range(len(flags)):
        pg.xbuf[0][r] = xbuf0[r].data_ptr()
        pg.xbuf[1][r] = xbuf1[r].data_ptr()
        pg.flags[r] = flags[r].data_ptr()
    pg.state = state.data_ptr()
    pg.world, pg.rank, pg.rows_max, pg.N = len(flags), rank, rows_max, N
    return pg


__all__ = ["PeerExchange", "PeerTimeout", "make_group", "DEFAULT_MAX_SPINS"]
