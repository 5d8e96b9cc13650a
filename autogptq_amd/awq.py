"""AWQ checkpoint ingest: the data-format step in front of the quantized-linear path.

Mirrors ``auto_gptq/modeling/_utils.py``:

* ``awq_reverse_reorder_int_tensor``  :525-553  -- undo AutoAWQ's nibble order along the last dimension
* ``unpack_awq``                      :556-621  -- AWQ words -> (fp16 weight [N, K], zero-points int8 [G, N])
* ``pack_from_tensors``               :624-701  -- (fp16 weight, zero-points, scales) -> GPTQ ``qweight`` / ``qzeros``

and adds ``repack_awq_to_gptq``: the composition of the last two as ONE integer pass on the GPU
(``gptq_awq_repack``), which is what a loader actually wants -- the reference's fp16 round trip exists only
because it reuses ``pack()``'s float interface.  4-bit only, as the reference (``assert bits == 4``).

The reference runs these on ``.cuda()`` tensors and returns GPU (unpack) / CPU (pack) tensors; here inputs may live
anywhere, the work runs on the current HIP device through libgptq_mi355x.so, results stay on that device.
There is no host implementation: without the library the calls raise.
"""
from __future__ import annotations

import torch

from . import _lib

_POS = (0, 4, 1, 5, 2, 6, 3, 7)      # column 8c + i lives at nibble _POS[i] of AWQ word c  (= order_map o order_map)


def _dev() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("autogptq_amd.awq needs a HIP device: there is no host implementation of the AWQ ingest kernels")
    return torch.device("cuda", torch.cuda.current_device())


def _check_bits(bits: int) -> None:
    if bits != 4:
        raise AssertionError("AWQ ingest is 4-bit only (auto_gptq/modeling/_utils.py:526,572,647)")


def awq_reverse_reorder_int_tensor(int_tensor: torch.Tensor, bits: int) -> torch.Tensor:
    """``out = int_tensor.T[:, reorder]`` with ``reorder`` = AutoAWQ's order map applied twice inside every run of 8
    columns (reference :533-553).  Pure indexing, runs on the tensor's own device."""
    _check_bits(bits)
    t = int_tensor.T.contiguous()
    if t.shape[-1] % 8:
        raise AssertionError("last dimension must be a multiple of 32 // bits")
    idx = (torch.arange(0, t.shape[-1], 8, device=t.device).reshape(-1, 1) +
           torch.tensor(_POS, device=t.device).reshape(1, -1)).reshape(-1)
    return t[:, idx]


def unpack_awq(awq_qweight: torch.Tensor, awq_qzeros: torch.Tensor, awq_scales: torch.Tensor, bits: int, group_size: int):
    """-> ``(fp16_weight [N, K] (transposed view of a [K, N] buffer, like the reference), zeros int8 [G, N])``."""
    _check_bits(bits)
    dev = _dev()
    K, NW = awq_qweight.shape
    N = NW * 8
    G = awq_qzeros.shape[0]
    if awq_scales.dtype != torch.float16:
        raise TypeError("awq_scales must be float16 (AutoAWQ checkpoints store fp16 scales)")
    if tuple(awq_qzeros.shape) != (G, NW) or tuple(awq_scales.shape) != (G, N) or G * group_size != K:
        raise ValueError(f"inconsistent AWQ shapes: qweight {tuple(awq_qweight.shape)}, qzeros {tuple(awq_qzeros.shape)}, "
                         f"scales {tuple(awq_scales.shape)}, group_size {group_size}")
    qw = awq_qweight.to(dev, torch.int32).contiguous()
    qz = awq_qzeros.to(dev, torch.int32).contiguous()
    sc = awq_scales.to(dev).contiguous()
    w_kn = torch.empty((K, N), dtype=torch.float16, device=dev)
    zeros = torch.empty((G, N), dtype=torch.int8, device=dev)
    lib = _lib.load()
    _lib.check(lib.gptq_awq_unpack(qw.data_ptr(), qz.data_ptr(), sc.data_ptr(), K, N, group_size, w_kn.data_ptr(), zeros.data_ptr(),
                                   _lib.current_stream_handle(dev)))
    return w_kn.T, zeros


def pack_from_tensors(unpacked_qweight: torch.Tensor, unpacked_qzeros: torch.Tensor, awq_scales: torch.Tensor, bits: int,
                      group_size: int):
    """(weight [N, K] float, zero-points [G, N] integer, scales [G, N]) -> GPTQ ``(qweight int32 [K/8, N], qzeros int32
    [G, N/8])`` with the reference's arithmetic: ``round((W + z*s) / s)`` in the tensors' dtype, zero field ``(z - 1) & 15``
    (reference :656-680).  Runs through gptq_pack_weights / gptq_pack_zeros."""
    _check_bits(bits)
    dev = _dev()
    N, K = unpacked_qweight.shape
    G = unpacked_qzeros.shape[0]
    if G * group_size != K or tuple(unpacked_qzeros.shape) != (G, N) or tuple(awq_scales.shape) != (G, N):
        raise ValueError("inconsistent shapes for pack_from_tensors")
    if unpacked_qweight.dtype not in _lib.DTYPE_ENUM or awq_scales.dtype not in _lib.DTYPE_ENUM:
        raise TypeError("weight / scales must be float16, bfloat16 or float32")
    W = unpacked_qweight.to(dev).contiguous()
    sc = awq_scales.to(dev).contiguous()
    z_int = unpacked_qzeros.to(dev, torch.int32)
    z_f = z_int.to(sc.dtype).contiguous()                        # z * s is formed in the scales' dtype, like int8 * half
    g_idx = (torch.arange(K, device=dev, dtype=torch.int32) // group_size).contiguous()
    qweight = torch.empty((K // 8, N), dtype=torch.int32, device=dev)
    qzeros = torch.empty((G, N // 8), dtype=torch.int32, device=dev)
    scales_out = torch.empty((G, N), dtype=W.dtype, device=dev)
    lib = _lib.load()
    st = _lib.current_stream_handle(dev)
    _lib.check(lib.gptq_pack_weights(W.data_ptr(), sc.data_ptr(), z_f.data_ptr(), g_idx.data_ptr(), K, N, 4, group_size,
                                     _lib.DTYPE_ENUM[W.dtype], _lib.DTYPE_ENUM[sc.dtype], qweight.data_ptr(), scales_out.data_ptr(), st))
    # pack() ORs the unmasked zero-1 (an all-ones word for z = 0); pack_from_tensors masks the field first: hand the
    # zero kernel z' = ((z - 1) & 15) + 1 so that its own "- 1" lands on the masked value
    zm = (((z_int - 1) & 15) + 1).to(torch.float32).contiguous()
    _lib.check(lib.gptq_pack_zeros(zm.data_ptr(), G, N, 4, _lib.GPTQ_F32, qzeros.data_ptr(), st))
    return qweight, qzeros


def repack_awq_to_gptq(awq_qweight: torch.Tensor, awq_qzeros: torch.Tensor, group_size: int):
    """AWQ ``(qweight [K, N/8], qzeros [G, N/8])`` -> GPTQ ``(qweight [K/8, N], qzeros [G, N/8])`` in one integer pass
    (``scales`` are shared by the two formats).  Equals ``pack_from_tensors(*unpack_awq(...))`` bit for bit."""
    dev = _dev()
    K, NW = awq_qweight.shape
    N = NW * 8
    G = awq_qzeros.shape[0]
    if tuple(awq_qzeros.shape) != (G, NW) or G * group_size != K:
        raise ValueError("inconsistent AWQ shapes")
    qw = awq_qweight.to(dev, torch.int32).contiguous()
    qz = awq_qzeros.to(dev, torch.int32).contiguous()
    qweight = torch.empty((K // 8, N), dtype=torch.int32, device=dev)
    qzeros = torch.empty((G, NW), dtype=torch.int32, device=dev)
    lib = _lib.load()
    _lib.check(lib.gptq_awq_repack(qw.data_ptr(), qz.data_ptr(), K, N, group_size, qweight.data_ptr(), qzeros.data_ptr(),
                                   _lib.current_stream_handle(dev)))
    return qweight, qzeros
