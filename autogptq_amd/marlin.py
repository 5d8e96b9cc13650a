"""Marlin-format checkpoints <-> the GPTQ v1 tensors this backend reads (a load-time data-format step).

The reference serialises Marlin layers as ``B int32 [K/16, 2N]`` (16x16 tiles, a 1024-entry permutation inside every run of
four tiles, nibbles interleaved 0,2,4,6,1,3,5,7) and ``s fp16 [G, N]`` (columns permuted inside runs of 64, or of 32 when the
layer has a single group): ``auto_gptq/nn_modules/qlinear/qlinear_marlin.py:51-80`` (tables) and ``:133-176`` (``pack``);
``autogptq_extension/marlin/marlin_repack.cu:8-92`` is the GPTQ -> Marlin direction on the GPU.  Marlin is symmetric int4: the
stored nibble is ``round(w / s) + 8``, i.e. a GPTQ layer whose zero-point is 8 everywhere (stored field 7).

Both directions here are pure index arithmetic on whole tensors (``torch`` gathers on whatever device the tensors live on --
a one-time pass over K*N/2 bytes at load time, not part of the per-token path), bit-exact against what the reference's own
``pack`` produced (``tests/golden/marlin_*.npz``).
"""
from __future__ import annotations

import torch

_INTERLEAVE = (0, 2, 4, 6, 1, 3, 5, 7)


def marlin_perms(device=None):
    """``(perm [1024], scale_perm [64], scale_perm_single [32])`` as int64 tensors (reference ``_get_perms``, :51-80)."""
    i = torch.arange(32, device=device)
    col, r = i // 4, i % 4
    rows = torch.stack([2 * r, 2 * r + 1, 2 * r + 8, 2 * r + 9], dim=-1)                               # [32, 4]
    p1 = (16 * rows[:, None, :] + col[:, None, None] + 8 * torch.arange(2, device=device)[None, :, None]).reshape(32, 8)
    perm = (p1[:, None, :] + 256 * torch.arange(4, device=device)[None, :, None]).reshape(-1, 8)
    perm = perm[:, torch.tensor(_INTERLEAVE, device=device)].reshape(-1)
    scale_perm = (torch.arange(8, device=device)[:, None] + 8 * torch.arange(8, device=device)[None, :]).reshape(-1)
    single = (2 * torch.arange(4, device=device)[:, None] +
              torch.tensor([0, 1, 8, 9, 16, 17, 24, 25], device=device)[None, :]).reshape(-1)
    return perm, scale_perm, single


def _check(K: int, N: int, group_size: int) -> int:
    gs = K if group_size == -1 else group_size
    if K % 128 or N % 256:
        raise ValueError("`infeatures` must be divisible by 128 and `outfeatures` by 256.")           # qlinear_marlin.py:96-97
    if gs not in (128, K):
        raise ValueError("Only group_size -1 and 128 are supported.")                                   # :100-101
    return gs


def marlin_to_gptq(B: torch.Tensor, s: torch.Tensor, group_size: int):
    """Marlin ``(B, s)`` -> GPTQ ``(qweight int32 [K/8, N], qzeros int32 [G, N/8], scales fp16 [G, N])``; read the result
    with either zero convention (the stored field 7 is not the wrapping value)."""
    K, N = B.shape[0] * 16, B.shape[1] // 2
    gs = _check(K, N, group_size)
    dev = B.device
    perm, sp, sps = marlin_perms(dev)
    shifts = 4 * torch.arange(8, device=dev, dtype=torch.int32)
    t = ((B.to(torch.int32).unsqueeze(-1) >> shifts) & 15).reshape(K // 16, N * 16)                    # nibble i of word j = element 8j + i
    t = t.reshape(-1, perm.numel())[:, torch.argsort(perm)]
    w = t.reshape(K // 16, N // 16, 16, 16).permute(0, 2, 1, 3).reshape(K, N)                          # [K, N], values w + 8 in 0..15
    sel = sp if gs != K else sps
    scales = s.reshape(-1, sel.numel())[:, torch.argsort(sel)].reshape(-1, N).contiguous()
    wq = w.reshape(K // 8, 8, N).to(torch.int64)
    qweight = torch.zeros((K // 8, N), dtype=torch.int64, device=dev)
    for j in range(8):
        qweight |= wq[:, j] << (4 * j)
    qweight = _to_i32(qweight)
    qzeros = torch.full((K // gs, N // 8), 0x77777777, dtype=torch.int32, device=dev)
    return qweight.contiguous(), qzeros, scales


def _to_i32(v64: torch.Tensor) -> torch.Tensor:
    v = v64 & 0xFFFFFFFF
    return torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32)


def gptq_to_marlin(qweight: torch.Tensor, qzeros: torch.Tensor, scales: torch.Tensor, group_size: int):
    """GPTQ ``(qweight, qzeros, scales)`` of a symmetric 4-bit layer (every zero field 7, sequential groups) -> Marlin
    ``(B int32 [K/16, 2N], s fp16 [G, N])`` -- the job of ``gptq_repack`` + the scale shuffle of ``pack``."""
    K, N = qweight.shape[0] * 8, qweight.shape[1]
    gs = _check(K, N, group_size)
    if not bool((qzeros.to(torch.int32) == 0x77777777).all()):
        raise ValueError("Marlin is symmetric: every stored zero field must be 7 (zero-point 8)")
    dev = qweight.device
    perm, sp, sps = marlin_perms(dev)
    shifts = 4 * torch.arange(8, device=dev, dtype=torch.int32)
    w = ((qweight.to(torch.int32).unsqueeze(1) >> shifts[None, :, None]) & 15).reshape(K, N)            # [K, N]
    t = w.reshape(K // 16, 16, N // 16, 16).permute(0, 2, 1, 3).reshape(K // 16, N * 16)
    t = t.reshape(-1, perm.numel())[:, perm].reshape(K // 16, N * 2, 8).to(torch.int64)
    B = torch.zeros((K // 16, N * 2), dtype=torch.int64, device=dev)
    for i in range(8):
        B |= t[:, :, i] << (4 * i)
    sel = sp if gs != K else sps
    s = scales.reshape(-1, sel.numel())[:, sel].reshape(-1, N).contiguous()
    return _to_i32(B).contiguous(), s
