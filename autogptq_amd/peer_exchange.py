"""Direct peer-store all-gather for column-parallel layers; experimental, off by default.

``PeerExchange`` owns what ``gptq_peer_group_t`` (include/gptq_mi355x.h) describes: per rank two exchange buffers
``[rows_max, N]``, the arrival flags and the 4-word state, plus the peers' buffers and flags mapped into this process.
PyTorch does the plumbing only -- device memory, and its CUDA-IPC tensor sharing (``torch.multiprocessing.reductions``)
to map a peer's allocation -- the exchange itself is the two launches of ``csrc/peer.hip`` through the C ABI.

The reference has nothing to mirror here: it runs a layer on one GPU (tests/test_q4.py:1224-1226 ``test_multigpu`` is a
TODO).  What was exercised: ranks sharing ONE device (in-process groups and 2 processes through IPC, tests/).  Across
GPUs the flags and buffers have to be FINE-GRAINED allocations for the system-scope fences of csrc/peer.hip to order
anything; the caching allocator only hands out coarse-grained memory, so by default the buffers come from
``hipExtMallocWithFlags(hipDeviceMallocFinegrained)`` (``FineGrainedBuffer``: ctypes on libamdhip64, wrapped as tensors
through ``__cuda_array_interface__``, shared between processes with ``hipIpcGetMemHandle`` / ``hipIpcOpenMemHandle``).
A group that spans several devices refuses coarse-grained buffers unless ``allow_coarse=True`` says otherwise.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from . import _lib

DEFAULT_MAX_SPINS = 1 << 22          # bounded wait: ~seconds of polling, then state[3] is raised instead of hanging the queue


_HIP = None


class _IpcHandle(ctypes.Structure):          # hipIpcMemHandle_t: 64 opaque bytes, passed BY VALUE to hipIpcOpenMemHandle
    _fields_ = [("reserved", ctypes.c_char * 64)]


def _hip():
    """libamdhip64 through ctypes (the runtime torch itself is linked against: same device context, same streams)."""
    global _HIP
    if _HIP is None:
        for name in ("libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"):
            try:
                _HIP = ctypes.CDLL(name)
                break
            except OSError:
                continue
        if _HIP is None:
            raise OSError("libamdhip64.so not found")
        _HIP.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        _HIP.hipFree.argtypes = [ctypes.c_void_p]
        _HIP.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        _HIP.hipIpcGetMemHandle.argtypes = [ctypes.POINTER(_IpcHandle), ctypes.c_void_p]
        _HIP.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), _IpcHandle, ctypes.c_uint]
        _HIP.hipIpcCloseMemHandle.argtypes = [ctypes.c_void_p]
    return _HIP


_TYPESTR = {torch.float16: "<f2", torch.bfloat16: "<u2", torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}


class _RawDeviceArray:
    """Just enough of __cuda_array_interface__ for torch.as_tensor to alias a raw device pointer (no copy, no ownership)."""

    def __init__(self, ptr: int, shape, typestr: str):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class FineGrainedBuffer:
    """One fine-grained device allocation (hipExtMallocWithFlags, hipDeviceMallocFinegrained = 0x1) or the mapping of a peer's:
    coherent at system scope, which is what stores + flags crossing xGMI need.  Owns the pointer; ``tensor()`` aliases it."""

    FINEGRAINED = 0x1

    def __init__(self, nbytes: int = 0, device=None, ipc_handle: bytes = None):
        hip = _hip()
        self.device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
        self.nbytes = int(nbytes)
        self._ptr = ctypes.c_void_p()
        self._mapped = ipc_handle is not None
        with torch.cuda.device(self.device):
            if self._mapped:
                h = _IpcHandle.from_buffer_copy(ipc_handle)
                rc = hip.hipIpcOpenMemHandle(ctypes.byref(self._ptr), h, 1)      # hipIpcMemLazyEnablePeerAccess
                if rc != 0:
                    raise RuntimeError(f"hipIpcOpenMemHandle failed ({rc})")
            else:
                rc = hip.hipExtMallocWithFlags(ctypes.byref(self._ptr), max(self.nbytes, 256), self.FINEGRAINED)
                if rc != 0:
                    raise RuntimeError(f"hipExtMallocWithFlags(finegrained) failed ({rc})")
                hip.hipMemset(self._ptr, 0, max(self.nbytes, 256))
                torch.cuda.synchronize(self.device)

    @property
    def ptr(self) -> int:
        return int(self._ptr.value)

    def tensor(self, shape, dtype: torch.dtype) -> torch.Tensor:
        if dtype == torch.bfloat16:                                              # no typestr for bf16: alias as u16 bits and view
            t = torch.as_tensor(_RawDeviceArray(self.ptr, shape, "<i2"), device=self.device)
            t = t.view(torch.bfloat16)
        else:
            t = torch.as_tensor(_RawDeviceArray(self.ptr, shape, _TYPESTR[dtype]), device=self.device)
        t._gptq_owner = self                                                     # the allocation lives as long as a tensor on it does
        return t

    def ipc_handle(self) -> bytes:
        h = _IpcHandle()
        with torch.cuda.device(self.device):
            rc = _hip().hipIpcGetMemHandle(ctypes.byref(h), self._ptr)
        if rc != 0:
            raise RuntimeError(f"hipIpcGetMemHandle failed ({rc})")
        return bytes(ctypes.string_at(ctypes.addressof(h), 64))

    def __del__(self):
        try:
            if self._ptr.value:
                if self._mapped:
                    _hip().hipIpcCloseMemHandle(self._ptr)
                else:
                    _hip().hipFree(self._ptr)
                self._ptr = ctypes.c_void_p()
        except Exception:            # interpreter shutdown
            pass


class PeerTimeout(RuntimeError):
    """A collect gave up waiting for a peer's slice (``state[3]`` raised by the kernel)."""


def _host_id() -> str:
    import socket
    return socket.gethostname()


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
                 buffers: Optional[dict] = None, max_spins: int = DEFAULT_MAX_SPINS, fine_grained: Optional[bool] = None,
                 allow_coarse: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.world > _lib.PEER_MAX:
            raise ValueError(f"peer-store exchange supports up to {_lib.PEER_MAX} ranks, got {self.world}")
        self.rows_max, self.N, self.dtype, self.device = rows_max, N, dtype, torch.device(device)
        self.max_spins = max_spins
        # which physical devices does the group span?  (system-scope fences only order fine-grained memory across GPUs)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        me = (_host_id(), idx)
        if self.world > 1:
            everyone_dev: List = [None] * self.world
            dist.all_gather_object(everyone_dev, me, group=group)
            self.multi_device = len(set(everyone_dev)) > 1
        else:
            self.multi_device = False
        self.fine_grained = False
        fg = None
        if buffers is None and fine_grained is not False:
            try:
                esz = torch.empty(0, dtype=dtype).element_size()
                fg = dict(xbuf0=FineGrainedBuffer(rows_max * N * esz, self.device), xbuf1=FineGrainedBuffer(rows_max * N * esz, self.device),
                          flags=FineGrainedBuffer(_lib.PEER_MAX * 4, self.device))
                buffers = dict(xbuf0=fg["xbuf0"].tensor((rows_max, N), dtype), xbuf1=fg["xbuf1"].tensor((rows_max, N), dtype),
                               flags=fg["flags"].tensor((_lib.PEER_MAX,), torch.int32))
                self.fine_grained = True
            except Exception as e:                       # no libamdhip64 / allocation refused: coarse-grained memory, one device only
                if fine_grained:
                    raise
                fg = None
                self._fg_error = repr(e)
        if buffers is None:
            buffers = dict(xbuf0=torch.empty((rows_max, N), dtype=dtype, device=device),
                           xbuf1=torch.empty((rows_max, N), dtype=dtype, device=device),
                           flags=torch.zeros(_lib.PEER_MAX, dtype=torch.int32, device=device))
        elif not self.fine_grained:
            self.fine_grained = bool(buffers.get("fine_grained", False))      # caller-provided buffers may say what they are
            buffers = {k: v for k, v in buffers.items() if k != "fine_grained"}
        if self.multi_device and not self.fine_grained and not allow_coarse:
            raise RuntimeError("PeerExchange: the group spans several GPUs but the exchange buffers are coarse-grained (caching-allocator) memory: "
                               "the system-scope fences of the exchange do not order it.  Let PeerExchange allocate (fine_grained=True), pass "
                               "fine-grained buffers, or allow_coarse=True to try anyway." + (f"  ({getattr(self, '_fg_error', '')})" if fg is None else ""))
        self.state = torch.zeros(4, dtype=torch.int32, device=device)
        self._own = buffers
        self._fg = fg
        torch.cuda.synchronize(self.device)
        if self.world == 1:
            mapped = [buffers]
        elif fg is not None:
            mine = {k: v.ipc_handle() for k, v in fg.items()}
            everyone: List = [None] * self.world
            dist.all_gather_object(everyone, mine, group=group)
            esz = torch.empty(0, dtype=dtype).element_size()
            mapped, self._peer_fg = [], []
            for r in range(self.world):
                if r == self.rank:
                    mapped.append(buffers)
                    continue
                pf = {k: FineGrainedBuffer(device=self.device, ipc_handle=h) for k, h in everyone[r].items()}
                self._peer_fg.append(pf)
                mapped.append(dict(xbuf0=pf["xbuf0"].tensor((rows_max, N), dtype), xbuf1=pf["xbuf1"].tensor((rows_max, N), dtype),
                                   flags=pf["flags"].tensor((_lib.PEER_MAX,), torch.int32)))
        else:
            mine = {k: _share(v) for k, v in buffers.items()}
            everyone = [None] * self.world
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

    def fused_ok(self, qlinear, M: int) -> bool:
        """Can ``forward_gather`` run this shard?  Mirrors gptq_forward_scatter's own preconditions (csrc/capi.hip): the scatter is the epilogue of the
        decode-copy kernel -- M <= 4, a PLAIN (no act-order: the gather-through-perm form has no scatter epilogue) 3/4/8-bit layer with its copy."""
        if qlinear._layer is None:
            qlinear.post_init()
        return (M <= 4 and getattr(qlinear, "_qweight_tiled", None) is not None and getattr(qlinear, "epilogue", "none") == "none"
                and not qlinear.act_order and qlinear.outfeatures * self.world == self.N and qlinear.scales.dtype == self.dtype)

    def forward_gather(self, qlinear, x2: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x2 [M, K] -> [M, N]: the rank's column shard ``qlinear`` (an mi355x QuantLinear) computes its slice and stores it into every rank's
        exchange buffer from its own epilogue (gptq_forward_scatter), then ONE collect launch.  Two launches per tensor-parallel layer."""
        from .qlinear_mi355x import exchange_tick, reserve_workspace

        if qlinear._layer is None:
            qlinear.post_init()
        M = x2.shape[0]
        if out is None:
            out = torch.empty((M, self.N), dtype=self.dtype, device=self.device)
        lib = _lib.load()
        need = int(lib.gptq_workspace_bytes(ctypes.byref(qlinear._layer), M))
        ws_ptr, ws_bytes = None, 0
        if need:
            buf = reserve_workspace(self.device, need)
            ws_ptr, ws_bytes = buf.data_ptr(), buf.numel()
            exchange_tick(self.device)
        st = _lib.current_stream_handle(self.device)
        _lib.check(lib.gptq_forward_gather(ctypes.byref(qlinear._layer), x2.data_ptr(), out.data_ptr(), M, ctypes.byref(self.pg), self.max_spins,
                                           ws_ptr, ws_bytes, st))
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


__all__ = ["PeerExchange", "PeerTimeout", "FineGrainedBuffer", "make_group", "DEFAULT_MAX_SPINS"]
