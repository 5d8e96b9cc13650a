"""``QuantLinear`` backend ``QUANT_TYPE = "mi355x"``: AutoGPTQ's quantized linear on MI355X (gfx950).

Mirrors the plugin surface of the reference's backends so it drops into
``auto_gptq.nn_modules.qlinear`` as just another one:

* constructor signature, attributes and buffers (``qweight / qzeros / scales / g_idx / bias``, same
  names, shapes, dtypes = the checkpoint ABI)    auto_gptq/nn_modules/qlinear/qlinear_cuda.py:27-103,
                                                  qlinear_cuda_old.py:26-105
* ``pack(linear, scales, zeros, g_idx)``          qlinear_cuda.py:108-203
* ``post_init()``                                 qlinear_exllama.py:106-119 (derive side buffers on device)
* ``forward(x)``                                  qlinear_cuda_old.py:202-355, qlinear_cuda.py:205-317

The matmul itself always runs in the HIP library (``libgptq_mi355x.so`` through the C ABI of
``include/gptq_mi355x.h``).  There is no PyTorch/CPU fallback: ``forward`` on a non-GPU tensor or
without the built library raises.
"""
from __future__ import annotations

import ctypes
import math
from logging import getLogger

import numpy as np
import torch
import torch.nn as nn

from . import _lib

logger = getLogger(__name__)

_WARNED = set()


def _warn_once(msg: str) -> None:
    if msg not in _WARNED:
        _WARNED.add(msg)
        logger.warning(msg)


# Scratch (split-K slabs, permuted x): one buffer per (device, stream), shared by every layer that runs on that stream (the
# reference keeps the same kind of per-device scratch in model.device_to_buffers, auto_gptq/modeling/_utils.py:448-470).
#   * keyed by stream: layers running concurrently on different streams never share split-K partials;
#   * a buffer that is outgrown is RETIRED, never freed: a hipGraph captured earlier has its address baked in, and memory
#     handed back to the caching allocator would be overwritten by later replays;
#   * a buffer allocated while its stream is capturing belongs to that graph's private pool: it serves that capture only and
#     is not reused by eager calls that later land on the same stream handle.
_WORKSPACE: dict = {}      # (device index, stream handle) -> (tensor, allocated_during_capture)
_RETIRED: list = []


def reserve_workspace(device, nbytes: int, stream=None) -> torch.Tensor:
    """Make sure the scratch of (device, stream) holds ``nbytes`` (stream: a handle, default = the current stream).
    Call with the largest need before hipGraph capture (``autogptq_post_init`` does) so that forward never allocates."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    device = torch.device("cuda", idx)
    if stream is None:
        stream = torch.cuda.current_stream(device).cuda_stream
    capturing = torch.cuda.is_current_stream_capturing()
    key = (idx, int(stream))
    ent = _WORKSPACE.get(key)
    if ent is not None:
        buf, was_captured = ent
        if buf.numel() >= nbytes and (capturing or not was_captured):
            return buf
        _RETIRED.append(buf)
        nbytes = max(int(nbytes), 2 * buf.numel() if buf.numel() < nbytes else buf.numel())
    # zeroed: the front of the buffer holds the arrival tickets of the in-launch K-split combine (gptq_mi355x.h)
    buf = torch.zeros(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
    _WORKSPACE[key] = (buf, capturing)
    return buf


try:                                   # raw hipStream_t of the current stream without building a torch.cuda.Stream object
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                 # pragma: no cover
    def _raw_stream(idx):
        return torch.cuda.current_stream(idx).cuda_stream


_ROWS_SCRATCH: dict = {}    # device index -> uint8 tensor: packed rows rebuilt from a decode copy (layers whose checkpoint layout was released)


def reserve_rows_scratch(device, nbytes: int) -> torch.Tensor:
    """One buffer per device, grown (never shrunk) to the largest released layer: gptq_unprepack_decode writes a layer's packed rows here right before a kernel
    that reads rows runs.  Calls on one stream are ordered; layers that run concurrently on several streams must not be released."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    buf = _ROWS_SCRATCH.get(idx)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=torch.device("cuda", idx))
        if idx in _ROWS_SCRATCH:
            _RETIRED.append(_ROWS_SCRATCH[idx])       # a captured graph may still hold the old address
        _ROWS_SCRATCH[idx] = buf
    return buf


def _is_sequential_g_idx(g_idx: torch.Tensor, group_size: int) -> bool:
    k = g_idx.numel()
    ref = torch.arange(k, dtype=torch.int64, device=g_idx.device) // group_size
    return bool(torch.equal(g_idx.to(torch.int64), ref))


class QuantLinear(nn.Module):
    QUANT_TYPE = "mi355x"
    EXCHANGE_CHECK_EVERY = 4096   # workspace-taking calls between two reads of the exchanges' sticky error word (0 = never; exchange_error() reads it on demand)
    _exchange_calls = 0
    TILED_DECODE = True       # post_init derives the strip-major side copy of qweight for the decode kernels (a second copy of the packed weights)
    # Round 5: ONE copy of the weights on the device.  With the switch on (or post_init(release_checkpoint_layout=True) / autogptq_post_init(...,
    # release_checkpoint_layout=True)) a PLAIN layer that carries its decode copy moves ``qweight`` to pinned host memory behind post_init -- state_dict()
    # still returns it, bit for bit -- and the decode and prefill kernels (M <= 4, 5..8 where planned, M >= ~512) run from the copy alone.  Row counts whose
    # kernel reads packed ROWS (the batched-decode band) rebuild them per call into a scratch shared by all layers of the device (gptq_unprepack_decode: the
    # exact inverse, ~10 us for a 4096 x 11008 layer).  Act-order layers keep their buffers (the copy is made of their re-sequenced rows).
    RELEASE_CHECKPOINT_LAYOUT = False
    _extra_bytes_logged = False

    def __init__(
        self,
        bits,
        group_size,
        infeatures,
        outfeatures,
        bias,
        use_cuda_fp16=True,
        kernel_switch_threshold=128,
        trainable=False,
        weight_dtype=torch.float16,
        zero_mode="auto",
        epilogue="none",
        **kwargs,
    ):
        super().__init__()
        if bits not in [2, 3, 4, 8]:
            raise NotImplementedError("Only 2,3,4,8 bits are supported.")
        if trainable:
            raise NotImplementedError("The mi355x QuantLinear backend is inference-only (trainable=True is not supported).")
        if infeatures % 32 != 0 or outfeatures % 32 != 0:
            raise ValueError("infeatures and outfeatures must be divisible by 32 (packed-layout requirement).")
        if weight_dtype not in _lib.DTYPE_ENUM:
            raise ValueError(f"weight_dtype must be float16, bfloat16 or float32, got {weight_dtype}")
        if zero_mode not in ("auto", "wrap", "nowrap"):
            raise ValueError("zero_mode must be 'auto', 'wrap' or 'nowrap'")
        if epilogue not in ("none", "silu_mul"):
            raise ValueError("epilogue must be 'none' or 'silu_mul'")
        if epilogue == "silu_mul" and outfeatures % 64 != 0:
            raise ValueError("epilogue='silu_mul' needs outfeatures divisible by 64 (columns are [gate | up])")

        self.infeatures = infeatures
        self.outfeatures = outfeatures
        self.bits = bits
        self.group_size = group_size if group_size != -1 else infeatures
        self.maxq = 2 ** self.bits - 1
        self.trainable = trainable
        self.use_cuda_fp16 = use_cuda_fp16            # accepted for make_quant compatibility; unused
        self.kernel_switch_threshold = kernel_switch_threshold
        self.zero_mode = zero_mode
        # 'silu_mul': the layer holds [gate | up] and forward returns silu(gate(x)) * up(x), [.., outfeatures // 2]
        self.epilogue = epilogue

        G = math.ceil(infeatures / self.group_size)
        self.register_buffer("qweight", torch.zeros((infeatures // 32 * bits, outfeatures), dtype=torch.int32))
        self.register_buffer("qzeros", torch.zeros((G, outfeatures // 32 * bits), dtype=torch.int32))
        self.register_buffer("scales", torch.zeros((G, outfeatures), dtype=weight_dtype))
        self.register_buffer(
            "g_idx", torch.tensor([i // self.group_size for i in range(infeatures)], dtype=torch.int32))
        if bias:
            self.register_buffer("bias", torch.zeros((outfeatures), dtype=weight_dtype))
        else:
            self.bias = None

        # derived, non-persistent state (never part of state_dict; checkpoint tensors stay intact)
        self._layer = None            # ctypes GptqLayer
        self._keepalive = ()          # tensors the raw pointers in _layer refer to
        self._qweight_tiled = self._qconst_tiled = None    # the decode copy (post_init), never part of state_dict
        self._released = False        # post_init(release_checkpoint_layout=True): qweight lives in pinned host memory, rows are rebuilt on demand
        self._ws_need = {}            # M -> workspace bytes
        self._ws0_mask = 0            # bit M set: M rows (1..63) are known to need no workspace -- what the C++ fast path (cext/fastfwd.cpp) serves; 0 = never
        self.act_order = None         # resolved by post_init

    # ------------------------------------------------------------------ state handling
    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._invalidate()
        return out

    def _invalidate(self):
        if self._layer is not None and _MULTI:
            # forward_multi caches ctypes argument arrays keyed by the layer structs they point at: drop every group this
            # layer was part of (a .to() / reload builds new structs; stale entries would pin the old ones for ever)
            lid = id(self._layer)
            for key in [k for k in _MULTI if lid in k]:
                del _MULTI[key]
        self._layer = None
        self._keepalive = ()
        self._qweight_tiled = self._qconst_tiled = None
        self._ws_need = {}
        self._ws0_mask = 0
        self._rows_need = {}
        object.__setattr__(self, "_parts", None)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._invalidate()

    def resolved_zero_mode(self) -> int:
        """'auto' = the convention of the reference class this backend stands in for: no act-order ->
        cuda_old (wrap, except its 3-bit branch), act-order -> cuda (no wrap). SURVEY App. B #1."""
        if self.zero_mode == "wrap":
            return _lib.ZERO_WRAP
        if self.zero_mode == "nowrap":
            return _lib.ZERO_NOWRAP
        if self.bits == 3 or self.act_order:
            return _lib.ZERO_NOWRAP
        return _lib.ZERO_WRAP

    # ------------------------------------------------------------------ post_init
    def post_init(self, temp_dq=None, tiled=None, release_checkpoint_layout=None):
        """Snapshot device pointers and, for act-order layers, derive the group-sorted copy of
        qweight plus the x permutation (side buffers; qweight itself is never modified -- the
        reference's exllama backends overwrite it in place, q4_matrix.cu:160)."""
        if self.qweight.device.type != "cuda" and self.scales.device.type == "cuda" and getattr(self, "_released", False):
            self.qweight = self.qweight.to(self.scales.device)      # a released layer initialised again: the rows come back first
        dev = self.qweight.device
        if dev.type != "cuda":
            raise RuntimeError("mi355x QuantLinear.post_init needs the module on a ROCm GPU device "
                               f"(got {dev}); there is no CPU path.")
        lib = _lib.load()
        if self.g_idx.numel() != self.infeatures:
            return self._post_init_fused_g_idx(temp_dq, tiled)
        # raw device pointers go to the kernels: every buffer has to live on the module's GPU (a host pointer would fault there)
        for name in ("qzeros", "scales", "g_idx", "bias"):
            t = getattr(self, name)
            if t is not None and t.device != dev:
                raise RuntimeError(f"mi355x QuantLinear.post_init: {name} is on {t.device} but qweight is on {dev}; move the "
                                   "whole module with .to(device) first")
        _lib.ensure_init(dev)
        for name in ("qweight", "qzeros", "scales", "g_idx"):
            t = getattr(self, name)
            if not t.is_contiguous():
                setattr(self, name, t.contiguous())
        if self.scales.dtype not in _lib.DTYPE_ENUM:
            raise ValueError(f"unsupported scales dtype {self.scales.dtype}")
        if self.bias is not None and self.bias.dtype != self.scales.dtype:
            self.bias = self.bias.to(self.scales.dtype)

        self.act_order = not _is_sequential_g_idx(self.g_idx, self.group_size)
        qweight_seq = perm = None
        g_idx_ptr = None
        if self.act_order:
            g_host = self.g_idx.to("cpu", torch.int32).contiguous()
            _lib.check(lib.gptq_validate_g_idx(g_host.data_ptr(), self.infeatures, self.scales.shape[0]))
            perm_host = torch.empty(self.infeatures, dtype=torch.int32)
            uniform = ctypes.c_int(0)
            _lib.check(lib.gptq_make_sequential(g_host.data_ptr(), self.infeatures, self.group_size,
                                                perm_host.data_ptr(), ctypes.byref(uniform)))
            g_idx_ptr = self.g_idx.data_ptr()
            if uniform.value:
                perm = perm_host.to(dev)
                qweight_seq = torch.empty_like(self.qweight)
                with torch.cuda.device(dev):
                    _lib.check(lib.gptq_resequence_qweight(self.qweight.data_ptr(), perm.data_ptr(), self.infeatures,
                                                           self.outfeatures, self.bits, qweight_seq.data_ptr(),
                                                           _lib.current_stream_handle(dev)))
        L = _lib.GptqLayer()
        L.qweight = self.qweight.data_ptr()
        L.qzeros = self.qzeros.data_ptr()
        L.scales = self.scales.data_ptr()
        L.g_idx = g_idx_ptr
        L.bias = _lib.ptr(self.bias)
        L.K, L.N, L.bits, L.group_size = self.infeatures, self.outfeatures, self.bits, self.group_size
        L.dtype = _lib.DTYPE_ENUM[self.scales.dtype]
        L.zero_mode = self.resolved_zero_mode()
        L.qweight_seq = _lib.ptr(qweight_seq)
        L.perm = _lib.ptr(perm)
        L.epilogue = _lib.EPI_SILU_MUL if self.epilogue == "silu_mul" else _lib.EPI_NONE
        L.qweight_tiled = L.qconst_tiled = None
        L.tiled_cols = 0
        # Decode copy (3-, 4- and 8-bit fp16 / bf16 layers; act-order ones from their re-sequenced rows -- the kernel gathers x through perm): what exllamav2's shuffle / Marlin's repack do at load time (q_matrix.cu:19-42,149;
        # marlin_repack.cu:8-92) -- into NON-PERSISTENT storage, the checkpoint tensors stay as they are.  Costs a second copy of the packed weights in
        # HBM; QuantLinear.TILED_DECODE = False (or post_init(tiled=False)) turns it off.
        qweight_tiled = qconst_tiled = None
        if tiled is None:
            tiled = self.TILED_DECODE
        if tiled:                                        # (every bit width since round 6: the C side decides which layers qualify)  act-order layers: a copy of the re-sequenced rows (qweight_seq above); [gate | up] layers with the fused epilogue: the C side decides
            tb, cb = ctypes.c_size_t(0), ctypes.c_size_t(0)
            if lib.gptq_prepack_decode_bytes(ctypes.byref(L), ctypes.byref(tb), ctypes.byref(cb)) == 0:      # a layer that does not qualify simply has none
                qweight_tiled = torch.empty(tb.value, dtype=torch.uint8, device=dev)
                qconst_tiled = torch.empty(cb.value, dtype=torch.uint8, device=dev)
                with torch.cuda.device(dev):
                    _lib.check(lib.gptq_prepack_decode(ctypes.byref(L), qweight_tiled.data_ptr(), qconst_tiled.data_ptr(), _lib.current_stream_handle(dev)))
                L.qweight_tiled, L.qconst_tiled, L.tiled_cols = qweight_tiled.data_ptr(), qconst_tiled.data_ptr(), _lib.STRIP_COLS
        self._layer = L
        self._layer_ref = ctypes.byref(L)
        self._layer_addr = ctypes.addressof(L)
        self._fwd = lib.gptq_forward_ex
        self._dev = dev
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()
        self._dev = torch.device("cuda", self._dev_index)
        self._w_dtype = self.scales.dtype
        self._n_out = self.outfeatures // 2 if self.epilogue == "silu_mul" else self.outfeatures
        self._keepalive = (self.qweight, self.qzeros, self.scales, self.g_idx, self.bias, qweight_seq, perm, qweight_tiled, qconst_tiled)
        self._qweight_tiled, self._qconst_tiled = qweight_tiled, qconst_tiled
        self._ws_need = {}
        self._ws0_mask = 0
        self._dt_code = _lib.fwd.dtype_code(self.scales) if _lib.fwd is not None else -1
        self._released = False
        self._rows_need = {}
        if release_checkpoint_layout is None:
            release_checkpoint_layout = self.RELEASE_CHECKPOINT_LAYOUT
        if qweight_tiled is not None:
            extra = qweight_tiled.numel() + qconst_tiled.numel() + (qweight_seq.numel() * 4 if qweight_seq is not None else 0)
            if release_checkpoint_layout and not self.act_order:
                # the checkpoint rows leave the HBM: the registered buffer becomes a pinned host tensor (state_dict() unchanged), the kernels that read rows
                # get them rebuilt into the shared scratch (_rows_scratch) right before they run
                self._qweight_rows_bytes = self.qweight.numel() * 4
                host = torch.empty(self.qweight.shape, dtype=self.qweight.dtype, pin_memory=True)
                host.copy_(self.qweight)
                self.qweight = host
                L.qweight = reserve_rows_scratch(dev, self._qweight_rows_bytes).data_ptr()
                self._keepalive = (None,) + tuple(self._keepalive[1:])
                self._released = True
            elif not QuantLinear._extra_bytes_logged:
                QuantLinear._extra_bytes_logged = True
                logger.info("mi355x QuantLinear: post_init keeps a decode copy of the packed weights next to the checkpoint tensors (+%d bytes for this layer; "
                            "%s).  QuantLinear.TILED_DECODE = False turns the copy off, RELEASE_CHECKPOINT_LAYOUT = True (or "
                            "autogptq_post_init(release_checkpoint_layout=True)) keeps ONE copy on the device.", extra,
                            "act-order: + the re-sequenced rows" if qweight_seq is not None else "plain layer: 2x the packed bytes")
        return self

    def _rebuild_rows(self, M: int, tuning=None) -> None:
        """Released layers: if the kernel planned for M rows reads packed ROWS, rebuild them from the decode copy into the device's shared scratch (whose address
        the layer struct already carries) -- in stream order right in front of the call.  The answer per row count is cached."""
        need = self._rows_need.get(M) if tuning is None else None
        if need is None:
            d = _lib.describe_plan(self._layer, M, tuning)
            need = not (d.get("kernel") in ("strips", "rows", "panel", "wide_sk", "wide_copy"))          # the kernels that read the decode copy
            if tuning is None:
                self._rows_need[M] = need
        if need:
            dev = self._dev
            buf = reserve_rows_scratch(dev, self._qweight_rows_bytes)
            if buf.data_ptr() != self._layer.qweight:
                self._layer.qweight = buf.data_ptr()
            with torch.cuda.device(dev):
                _lib.check(_lib.load().gptq_unprepack_decode(self._qweight_tiled.data_ptr(), self.infeatures, self.outfeatures, self.bits, buf.data_ptr(),
                                                             _lib.current_stream_handle(dev)))

    def _post_init_fused_g_idx(self, temp_dq, tiled):
        """``len(g_idx) == n * infeatures``: the reference's fused q/k/v module of act-order projections (fused_llama_attn.py:186 concatenates the three g_idx;
        qlinear_cuda.py:300-312 then dequantises column block i -- ``infeatures`` columns wide, its ``num_dim`` -- with g_idx[i K : (i + 1) K]).  Here the n
        column blocks become n internal layers, each with its own activation order (side copies of the column slices: the checkpoint tensors stay as they
        are and stay the module's state_dict), and forward runs them through ``forward_multi``: the same values as the reference's per-block loop."""
        K, N = self.infeatures, self.outfeatures
        n = self.g_idx.numel() // K if K else 0
        if n < 2 or self.g_idx.numel() != n * K or N != n * K:
            raise NotImplementedError(f"len(g_idx) = {self.g_idx.numel()} is neither infeatures ({K}) nor n * infeatures with outfeatures = n * infeatures "
                                      "(the reference's fused-QKV layout, qlinear_cuda.py:300-312).")
        if self.epilogue != "none":
            raise NotImplementedError("a fused-QKV g_idx cannot be combined with an output epilogue")
        zpw = self.bits * K // 32                       # qzeros words per column block
        parts = []
        for i in range(n):
            p = QuantLinear(self.bits, self.group_size, K, K, self.bias is not None, weight_dtype=self.scales.dtype, zero_mode=self.zero_mode)
            p.qweight = self.qweight[:, i * K:(i + 1) * K].contiguous()
            p.qzeros = self.qzeros[:, i * zpw:(i + 1) * zpw].contiguous()
            p.scales = self.scales[:, i * K:(i + 1) * K].contiguous()
            p.g_idx = self.g_idx[i * K:(i + 1) * K].to(torch.int32).contiguous()
            if self.bias is not None:
                p.bias = self.bias[i * K:(i + 1) * K].contiguous()
            p.post_init(temp_dq, tiled)
            parts.append(p)
        object.__setattr__(self, "_parts", parts)       # not registered submodules: derived state, never part of state_dict
        self.act_order = any(p.act_order for p in parts)
        return self

    # ------------------------------------------------------------------ forward
    def _workspace(self, M: int, device, tuning=None):
        need = self._ws_need.get(M) if tuning is None else None
        if need is None:
            need = int(_lib.load().gptq_workspace_bytes_ex(ctypes.byref(self._layer), M,
                                                            ctypes.byref(tuning) if tuning is not None else None))
            if tuning is None:
                self._ws_need[M] = need
                if need == 0 and 0 < M < 64 and _lib.fwd is not None:
                    self._ws0_mask |= 1 << M
        if need == 0:
            return None, 0
        buf = reserve_workspace(device, need)
        exchange_tick(device)
        return buf.data_ptr(), buf.numel()

    def forward(self, x: torch.Tensor, tuning: "_lib.GptqTuning | None" = None):
        # The reference's callers are eager (generate() under inference_mode, auto_gptq/modeling/_base.py:415-418), and a decode
        # kernel here runs for ~5 us: everything per call that is not the launch is kept to attribute reads -- device, dtype,
        # ctypes handles and the layer pointer are resolved once in post_init.  Row counts already known to need no workspace (decode rows: the mask is 0 until
        # the path below has seen the row count once) go through the C++ fast path: checks on x, at::empty, current stream and the C-ABI call in one
        # METH_FASTCALL entry (cext/fastfwd.cpp); it answers None for anything but the plain case.
        mask = self._ws0_mask
        if mask and tuning is None and not self._released:
            r = _lib.fwd.forward(self._layer_addr, x, self.infeatures, self._n_out, self._dt_code, self._dev_index, mask)
            if r is not None:
                if r.__class__ is int:
                    _lib.check(r)
                return r
        if self._layer is None:
            if getattr(self, "_parts", None) is None:
                if x.device.type != "cuda":
                    raise RuntimeError("mi355x QuantLinear.forward needs a ROCm GPU tensor "
                                       f"(got {x.device}); there is no CPU path in this backend.")
                self.post_init()
            if getattr(self, "_parts", None) is not None:      # fused-QKV g_idx: n column blocks with their own activation orders
                return torch.cat(forward_multi(self._parts, x, tuning), dim=-1)
        dev = self._dev
        if x.device != dev:
            if x.device.type != "cuda":
                raise RuntimeError("mi355x QuantLinear.forward needs a ROCm GPU tensor "
                                   f"(got {x.device}); there is no CPU path in this backend.")
            raise RuntimeError(f"mi355x QuantLinear.forward: input is on {x.device}, the layer on {dev}")
        K = self.infeatures
        if x.shape[-1] != K:
            raise RuntimeError(f"input has {x.shape[-1]} features, layer expects {K}")
        w_dtype = self._w_dtype
        x_dtype = x.dtype
        x2 = x
        if x_dtype != w_dtype:
            _warn_once(f"mi355x QuantLinear: activation dtype {x_dtype} != weight dtype {w_dtype}; casting the "
                       f"activation to {w_dtype} (the result is cast back).")
            x2 = x2.to(w_dtype)
        if x2.dim() != 2:
            x2 = x2.reshape(-1, K)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        n_out = self._n_out
        out = torch.empty((M, n_out), dtype=w_dtype, device=dev)
        if M != 0 and self._released:
            self._rebuild_rows(M, tuning)
        if M != 0:
            need = self._ws_need.get(M) if tuning is None else None
            if need == 0:
                ws_ptr, ws_bytes = None, 0
            else:
                ws_ptr, ws_bytes = self._workspace(M, dev, tuning)
            idx = self._dev_index
            fast = _lib.fast
            if fast is not None:           # METH_FASTCALL trampoline: integers only
                taddr = ctypes.addressof(tuning) if tuning is not None else 0
                if idx != torch.cuda.current_device():
                    with torch.cuda.device(idx):
                        rc = fast.forward(self._layer_addr, x2.data_ptr(), out.data_ptr(), M, ws_ptr or 0, ws_bytes, _raw_stream(idx), taddr)
                else:
                    rc = fast.forward(self._layer_addr, x2.data_ptr(), out.data_ptr(), M, ws_ptr or 0, ws_bytes, _raw_stream(idx), taddr)
            else:
                tref = ctypes.byref(tuning) if tuning is not None else None
                if idx != torch.cuda.current_device():
                    with torch.cuda.device(idx):
                        rc = self._fwd(self._layer_ref, x2.data_ptr(), out.data_ptr(), M, ws_ptr, ws_bytes, _raw_stream(idx), tref)
                else:
                    rc = self._fwd(self._layer_ref, x2.data_ptr(), out.data_ptr(), M, ws_ptr, ws_bytes, _raw_stream(idx), tref)
            if rc:
                _lib.check(rc)
        if x_dtype != w_dtype:
            out = out.to(x_dtype)
        if x.dim() != 2:
            out = out.reshape(x.shape[:-1] + (n_out,))
        return out

    # ------------------------------------------------------------------ dequant / unpack helpers
    def dequantize(self) -> torch.Tensor:
        """[K, N] dequantised weight in the scales dtype (bit-exact w.r.t. the reference's `weights`)."""
        if self._layer is None and getattr(self, "_parts", None) is None:
            self.post_init()
        if getattr(self, "_parts", None) is not None:
            return torch.cat([p.dequantize() for p in self._parts], dim=1)
        dev = self.scales.device
        if self._released:
            self._rows_need.pop(-1, None)
            self._rows_need[-1] = True
            buf = reserve_rows_scratch(dev, self._qweight_rows_bytes)
            self._layer.qweight = buf.data_ptr()
            with torch.cuda.device(dev):
                _lib.check(_lib.load().gptq_unprepack_decode(self._qweight_tiled.data_ptr(), self.infeatures, self.outfeatures, self.bits, buf.data_ptr(),
                                                             _lib.current_stream_handle(dev)))
        W = torch.empty((self.infeatures, self.outfeatures), dtype=self.scales.dtype, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().gptq_dequant(ctypes.byref(self._layer), W.data_ptr(),
                                                _lib.current_stream_handle(dev)))
        return W

    # ------------------------------------------------------------------ pack
    def pack(self, linear, scales, zeros, g_idx=None):
        """Quantise ``linear.weight`` with (scales, zeros) [N, G] and fill qweight/qzeros/scales/g_idx/bias.

        Same arithmetic order and dtype promotion as the reference (qlinear_cuda.py:117-131):
        ``round((W + zeros*scales) / scales.to(W.dtype))``, fields OR-ed unmasked.  Runs on the GPU
        through gptq_pack_weights/gptq_pack_zeros when one is present (the reference can only pack on
        CPU, auto_gptq/modeling/_utils.py:303-309), otherwise as vectorised host tensor ops; results
        are returned on the device the module lives on, as the reference leaves them on CPU.
        """
        import transformers

        W = linear.weight.data.clone()
        if isinstance(linear, nn.Conv2d):
            W = W.flatten(1)
        if isinstance(linear, transformers.pytorch_utils.Conv1D):
            W = W.t()
        if g_idx is not None:
            self.g_idx = g_idx.clone().to(torch.int32)
        gi = self.g_idx.to(torch.int64)
        home = self.qweight.device

        if not zeros.is_floating_point():
            # integer zero-points (tests/test_hpu_linear.py:140-150 hands over int32): zeros * scales promotes to the scales
            # dtype and (zeros - 1).astype(uint32) wraps -1 to 0xFFFFFFFF -- both identical after an exact cast to float
            zeros = zeros.to(scales.dtype)
        scales_t = scales.t().contiguous()
        zeros_t = zeros.t().contiguous()
        if linear.bias is not None:
            self.bias = linear.bias.clone().to(dtype=linear.weight.dtype)

        use_gpu = torch.cuda.is_available() and W.dtype in _lib.DTYPE_ENUM and scales_t.dtype in _lib.DTYPE_ENUM \
            and zeros_t.dtype == scales_t.dtype
        if use_gpu:
            qweight, qzeros, scales_out = self._pack_device(W, scales_t, zeros_t, gi)      # raises if the library is missing
        else:
            # No HIP device (or a dtype outside the C ABI): the reference's own pack() is CPU code, and this is its
            # vectorised equivalent, bit-identical (tests/test_host_logic.py::test_host_pack_bit_exact).  Said out loud,
            # because everything else in this backend refuses to run on the host.
            _warn_once("mi355x QuantLinear.pack: no HIP device available (or unsupported dtype) -- packing on the host, "
                       "as the reference does")
            qweight, qzeros, scales_out = self._pack_host(W, scales_t, zeros_t, gi)
        self.qweight = qweight.to(home)
        self.qzeros = qzeros.to(home)
        self.scales = scales_out.to(home)
        self.g_idx = self.g_idx.to(home)                     # every buffer ends up where the module lives
        if self.bias is not None:
            self.bias = self.bias.to(home)
        self._invalidate()

    def _pack_device(self, W, scales_t, zeros_t, gi):
        lib = _lib.load()
        dev = torch.device("cuda", torch.cuda.current_device())
        K, N, bits = self.infeatures, self.outfeatures, self.bits
        Wd = W.contiguous().to(dev)
        sd, zd = scales_t.to(dev), zeros_t.to(dev)
        gd = gi.to(dev, torch.int32).contiguous()
        qweight = torch.empty((K // 32 * bits, N), dtype=torch.int32, device=dev)
        qzeros = torch.empty((sd.shape[0], N // 32 * bits), dtype=torch.int32, device=dev)
        scales_out = torch.empty(sd.shape, dtype=W.dtype, device=dev)
        st = _lib.current_stream_handle(dev)
        _lib.check(lib.gptq_pack_weights(Wd.data_ptr(), sd.data_ptr(), zd.data_ptr(), gd.data_ptr(), K, N, bits,
                                         self.group_size, _lib.DTYPE_ENUM[W.dtype], _lib.DTYPE_ENUM[sd.dtype],
                                         qweight.data_ptr(), scales_out.data_ptr(), st))
        _lib.check(lib.gptq_pack_zeros(zd.data_ptr(), sd.shape[0], N, bits, _lib.DTYPE_ENUM[zd.dtype],
                                       qzeros.data_ptr(), st))
        torch.cuda.synchronize(dev)
        return qweight, qzeros, scales_out

    def _pack_host(self, W, scales_t, zeros_t, gi):
        bits = self.bits
        scale_zeros = zeros_t * scales_t
        s_cast = scales_t.clone().to(dtype=W.dtype)
        intweight = torch.round((W.t() + scale_zeros[gi]) / s_cast[gi]).to(torch.int)   # [K, N]
        qweight = _pack_fields(intweight.to(torch.int64) & 0xFFFFFFFF, bits)
        zm1 = (zeros_t - 1).to(torch.int64) & 0xFFFFFFFF                                 # [G, N]
        qzeros = _pack_fields(zm1.t().contiguous(), bits).t().contiguous()
        return qweight, qzeros, s_cast


def _pack_fields(vals_u32: torch.Tensor, bits: int) -> torch.Tensor:
    """Pack uint32 values (held in int64) along dim 0 into int32 words, OR-ing whole (unmasked)
    values exactly as the reference does (qlinear_cuda.py:139-162)."""
    mask = 0xFFFFFFFF
    V = vals_u32.shape[0]
    if bits in (2, 4, 8):
        per = 32 // bits
        v = vals_u32.reshape(V // per, per, *vals_u32.shape[1:])
        out = torch.zeros((V // per,) + tuple(vals_u32.shape[1:]), dtype=torch.int64)
        for j in range(per):
            out |= (v[:, j] << (bits * j)) & mask
    else:
        v = vals_u32.reshape(V // 32, 32, *vals_u32.shape[1:])
        w0 = torch.zeros((V // 32,) + tuple(vals_u32.shape[1:]), dtype=torch.int64)
        w1 = torch.zeros_like(w0)
        w2 = torch.zeros_like(w0)
        for j in range(10):
            w0 |= (v[:, j] << (3 * j)) & mask
        w0 |= (v[:, 10] << 30) & mask
        w1 |= (v[:, 10] >> 2) & 1
        for j in range(10):
            w1 |= (v[:, 11 + j] << (3 * j + 1)) & mask
        w1 |= (v[:, 21] << 31) & mask
        w2 |= (v[:, 21] >> 1) & 3
        for j in range(10):
            w2 |= (v[:, 22 + j] << (3 * j + 2)) & mask
        out = torch.stack([w0, w1, w2], dim=1).reshape((V // 32 * 3,) + tuple(vals_u32.shape[1:]))
    out = torch.where(out >= 2 ** 31, out - 2 ** 32, out)
    return out.to(torch.int32).contiguous()


def share_act_order(layers) -> bool:
    """Act-order layers that read the same input and carry the SAME g_idx -- q / k / v and gate / up of a GPTQ checkpoint do: the order comes from the
    Hessian of their common input (the reference's fused q/k/v caller relies on it, fused_llama_attn.py:188) -- are pointed at ONE ``perm`` buffer; the
    C ABI takes the shared pointer as "these layers read one permuted x" and permutes x once per gptq_forward_multi call instead of once per layer.
    Compared once (when forward_multi first sees the group); True if the group now shares."""
    a = layers[0]
    if len(layers) < 2 or a._layer is None or not a._layer.perm or not a._layer.qweight_seq:
        return False
    for l in layers[1:]:
        if l._layer is None or not l._layer.perm or l.g_idx.shape != a.g_idx.shape or l.g_idx.device != a.g_idx.device or not torch.equal(l.g_idx, a.g_idx):
            return False
    for l in layers[1:]:
        l._layer.perm = a._layer.perm
        l._keepalive = tuple(l._keepalive) + (a._keepalive,)          # layer 0's perm tensor lives as long as the sharers do
    return True


def forward_multi(layers, x: torch.Tensor, tuning: "_lib.GptqTuning | None" = None):
    """``[l(x) for l in layers]`` for mi355x QuantLinears that read the same input, through gptq_forward_multi: ONE launch for
    q/k/v or gate/up of a decode step (M <= 4; 3- / 4- / 8-bit layers that carry their decode copy, plain or act-order), batched-decode and prefill
    rows through the multi-layer kernels or layer by layer (act-order layers of one g_idx then share ONE permuted x).  The checkpoint tensors
    are used where they are (the reference's fused modules concatenate copies of them, fused_llama_attn.py:171-203)."""
    # Everything that depends only on the GROUP is checked and resolved once per group (keyed by the layers' C structs) and kept in _MULTI: a decode
    # launch here takes 4.5 - 12 us, and the reference's callers are eager (generate() under inference_mode) -- per call only what depends on x remains.
    a = layers[0]
    n = len(layers)
    released = any(getattr(l, "_released", False) for l in layers)
    if released and tuning is not None:
        # a forced kernel family may read the packed ROWS, which released layers only have in the one shared scratch (whichever layer was rebuilt last):
        # the grouped entry points refuse the override instead of returning another layer's weights (single calls rebuild the rows: QuantLinear.forward)
        raise RuntimeError("forward_multi: layers with release_checkpoint_layout=True take no tuning override in the grouped entry points; call them one by one")
    if released and x.numel() // max(1, a.infeatures) > 4:
        # layers whose checkpoint rows left the HBM share ONE rows scratch: beyond the decode rows (which run from the copies) they are called one by one
        return [l(x, tuning) for l in layers]
    key = tuple([id(l._layer) for l in layers]) if a._layer is not None else None
    ent = _MULTI.get(key) if key is not None else None
    if ent is None:
        for l in layers:
            if l._layer is None:
                l.post_init()
        if any(l.infeatures != a.infeatures for l in layers):
            raise RuntimeError("forward_multi: every layer must take the input's feature count")
        if any(l._w_dtype != a._w_dtype for l in layers):
            raise RuntimeError("forward_multi: the layers must share the weight dtype")
        if any(l._dev != a._dev for l in layers):
            raise RuntimeError("forward_multi: the layers must be on one device")
        share_act_order(layers)
        key = tuple([id(l._layer) for l in layers])
        arr = (ctypes.POINTER(_lib.GptqLayer) * n)(*[ctypes.pointer(l._layer) for l in layers])
        optr_arr = (ctypes.c_void_p * n)()
        ent = _MULTI[key] = (arr, {}, [l._layer for l in layers], optr_arr, ctypes.addressof(arr), ctypes.addressof(optr_arr),
                             a._dev, a.infeatures, a._w_dtype, tuple(l._n_out for l in layers), a._dev_index, sum(l._n_out for l in layers))
    arr, need_by_m, _, optrs, arr_addr, optr_addr, dev, K, w_dtype, n_outs, idx = ent[:11]
    if tuning is None:
        mask = need_by_m.get(-1)           # row counts of this group known to need no workspace: the C++ fast path (cext/fastfwd.cpp) serves them
        if mask:
            r = _lib.fwd.forward_multi(arr_addr, n, x, K, n_outs, a._dt_code, idx, optr_addr, mask)
            if r is not None:
                if r.__class__ is int:
                    _lib.check(r)
                return list(r)
    if x.device != dev:
        raise RuntimeError(f"mi355x forward_multi: input is on {x.device}, the layers on {dev}")
    if x.shape[-1] != K:
        raise RuntimeError("forward_multi: every layer must take the input's feature count")
    x_dtype = x.dtype
    x2 = x.to(w_dtype) if x_dtype != w_dtype else x
    if x2.dim() != 2:
        x2 = x2.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    if M == 1:      # one decode row: ONE allocation, the outputs are its column slices (contiguous for a single row) -- torch.empty is 1.9 us of a ~10 us call
        outs = list(torch.empty((1, ent[11]), dtype=w_dtype, device=dev).split(n_outs, dim=1))
    else:
        outs = [torch.empty((M, no), dtype=w_dtype, device=dev) for no in n_outs]
    if M:
        need = need_by_m.get(M) if tuning is None else None
        if need is None:
            tref = ctypes.byref(tuning) if tuning is not None else None
            need = int(_lib.load().gptq_workspace_bytes_multi_ex(arr, n, M, tref))
            if tuning is None:
                need_by_m[M] = need
                if need == 0 and 0 < M < 64 and _lib.fwd is not None:
                    need_by_m[-1] = need_by_m.get(-1, 0) | (1 << M)
        ws_ptr, ws_bytes = 0, 0
        if need:
            buf = reserve_workspace(dev, need)
            ws_ptr, ws_bytes = buf.data_ptr(), buf.numel()
            exchange_tick(dev)
        for i in range(n):
            optrs[i] = outs[i].data_ptr()
        fast = _lib.fast
        def launch():
            if fast is not None:
                return fast.forward_multi(arr_addr, n, x2.data_ptr(), optr_addr, M, ws_ptr, ws_bytes, _raw_stream(idx),
                                          ctypes.addressof(tuning) if tuning is not None else 0)
            return _lib.load().gptq_forward_multi_ex(arr, n, x2.data_ptr(), optrs, M, ws_ptr or None, ws_bytes, _raw_stream(idx),
                                                     ctypes.byref(tuning) if tuning is not None else None)
        if idx != torch.cuda.current_device():              # rare: the layers' device is not the current one
            with torch.cuda.device(idx):
                rc = launch()
        else:
            rc = launch()
        if rc:
            _lib.check(rc)
    if x_dtype != w_dtype:
        outs = [o.to(x_dtype) for o in outs]
    if x.dim() != 2:
        outs = [o.reshape(x.shape[:-1] + (o.shape[-1],)) for o in outs]
    return outs


_MULTI: dict = {}


def mlp_forward(gate: QuantLinear, up: QuantLinear, down: QuantLinear, x: torch.Tensor, tuning: "_lib.GptqTuning | None" = None):
    """``down(silu(gate(x)) * up(x))`` for three mi355x QuantLinears through ONE C-ABI call (gptq_mlp_forward): the role of the
    reference's FusedLlamaMLPForQuantizedModel.forward (auto_gptq/nn_modules/fused_llama_mlp.py:157-242).  gate and up run as one
    multi-layer launch for decode rows, the SiLU * mul on fp32, then down -- any bits / act-order / row count the single layers take,
    checkpoint tensors untouched, one Python -> C transition instead of three.  (Round 3's one-launch persistent kernel was measured slower and is a lab
    now -- tools/lab/mlp_ring.hip; ``tuning`` with a path override is refused.)"""
    for l in (gate, up, down):
        if l._layer is None:
            l.post_init()
    if any(getattr(l, "_released", False) for l in (gate, up, down)):
        if tuning is not None:
            raise RuntimeError("mlp_forward: layers with release_checkpoint_layout=True take no tuning override")
        if x.numel() // max(1, gate.infeatures) > 4:
            g, u = forward_multi([gate, up], x)          # released layers (one shared rows scratch): composed from single calls
            return down(torch.nn.functional.silu(g) * u)
    dev = gate._dev
    if x.device != dev:
        raise RuntimeError(f"mi355x mlp_forward: input is on {x.device}, the layers on {dev}")
    K = gate.infeatures
    if x.shape[-1] != K or up.infeatures != K or up.outfeatures != gate.outfeatures or down.infeatures != gate.outfeatures:
        raise RuntimeError("mlp_forward: gate / up must be [K -> I] on the input's K features and down [I -> N]")
    w_dtype = gate._w_dtype
    if up._w_dtype != w_dtype or down._w_dtype != w_dtype:
        raise RuntimeError("mlp_forward: the three layers must share the weight dtype")
    x_dtype = x.dtype
    x2 = x.to(w_dtype) if x_dtype != w_dtype else x
    if x2.dim() != 2:
        x2 = x2.reshape(-1, K)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    M = x2.shape[0]
    out = torch.empty((M, down.outfeatures), dtype=w_dtype, device=dev)
    if M:
        lib = _lib.load()
        key = (id(gate._layer), id(up._layer), id(down._layer), "mlp")
        need_by_m = _MULTI.get(key)
        if need_by_m is None:
            need_by_m = _MULTI[key] = {}
            share_act_order([gate, up])                 # act-order gate / up with one g_idx: one permuted x for both
        tref = ctypes.byref(tuning) if tuning is not None else None
        need = need_by_m.get(M) if tuning is None else None
        if need is None:
            need = int(lib.gptq_workspace_bytes_mlp_ex(gate._layer_ref, up._layer_ref, down._layer_ref, M, tref))
            if tuning is None:
                need_by_m[M] = need
        buf = reserve_workspace(dev, max(need, 1))
        exchange_tick(dev)
        idx = gate._dev_index
        fast = _lib.fast

        def launch():
            if fast is not None and tuning is None and hasattr(fast, "mlp_forward"):
                return fast.mlp_forward(gate._layer_addr, up._layer_addr, down._layer_addr, x2.data_ptr(), out.data_ptr(), M,
                                        buf.data_ptr(), buf.numel(), _raw_stream(idx))
            return lib.gptq_mlp_forward_ex(gate._layer_ref, up._layer_ref, down._layer_ref, x2.data_ptr(), out.data_ptr(), M,
                                           buf.data_ptr(), buf.numel(), _raw_stream(idx), tref)
        if idx != torch.cuda.current_device():
            with torch.cuda.device(idx):
                rc = launch()
        else:
            rc = launch()
        if rc:
            _lib.check(rc)
    if x_dtype != w_dtype:
        out = out.to(x_dtype)
    if x.dim() != 2:
        out = out.reshape(x.shape[:-1] + (out.shape[-1],))
    return out


def exchange_tick(device) -> None:
    """Called by every entry point that hands the kernels a workspace (QuantLinear.forward, forward_multi, mlp_forward, PeerExchange.forward_gather): those
    are the calls with an in-launch exchange (K slices, balanced tail, stream-K pieces).  Every QuantLinear.EXCHANGE_CHECK_EVERY of them the sticky error
    word of the workspace is read (a device -> host read: a sync point, hence periodic; never inside a stream capture).  A bounded wait that gave up has
    produced a wrong result somewhere since the last check -- it must not pass silently.  The word is cleared once reported, so ONE timeout raises once."""
    n = QuantLinear._exchange_calls = QuantLinear._exchange_calls + 1
    every = QuantLinear.EXCHANGE_CHECK_EVERY
    if every and n % every == 0 and not torch.cuda.is_current_stream_capturing():
        try:
            bad = exchange_error(device, clear=True)
        except RuntimeError:                    # e.g. another thread is capturing in global mode: a synchronising read is not allowed now -- next time
            bad = False
        if bad:
            raise RuntimeError("gptq_mi355x: a bounded wait of an in-launch exchange gave up (another client kept part of the GPU busy?): at least one result "
                               "since the last check is wrong; see exchange_error()")


def exchange_error(device=None, clear: bool = False) -> bool:
    """True if a BOUNDED wait of an in-launch exchange (the K-slice combines of the decode and 17..256-row kernels, the balanced tail of the tiled GEMM)
    ever gave up on this device's current-stream workspace: the sticky error word in the workspace header's tail (gptq_mi355x.h).  A launch whose wait
    gave up has produced a wrong result; the kernels never hang instead.  Synchronises the stream."""
    device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ent = _WORKSPACE.get((idx, int(torch.cuda.current_stream(device).cuda_stream)))
    if ent is None:
        return False
    torch.cuda.synchronize(device)
    tail = ent[0][_lib.WS_HEADER_BYTES - 64:_lib.WS_HEADER_BYTES].view(torch.int32)
    bad = bool(tail[2].item() != 0)
    if bad and clear:
        tail[2] = 0
    return bad


mlp_exchange_error = exchange_error      # round-3 name


__all__ = ["QuantLinear", "reserve_workspace", "forward_multi", "mlp_forward", "exchange_error", "exchange_tick", "mlp_exchange_error"]
