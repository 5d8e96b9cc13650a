"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the Marlin checkpoint layout (never imported by the product).

Follows auto_gptq/nn_modules/qlinear/qlinear_marlin.py: the permutation tables of _get_perms (:51-80) and the integer part of
QuantLinear.pack (:133-176).  Pinned by tests/golden/marlin_*.npz, which tests/golden/make_golden_marlin.py produced by
running the reference's own pack() (tests/test_oracle_golden.py).
"""
import numpy as np


def perms():
    """(perm [1024], scale_perm [64], scale_perm_single [32]) -- qlinear_marlin.py:51-80, written as index arithmetic."""
    i = np.arange(32)
    col, r = i // 4, i % 4
    rows = np.stack([2 * r, 2 * r + 1, 2 * r + 8, 2 * r + 9], axis=-1)                       # [32, 4]
    p1 = (16 * rows[:, None, :] + col[:, None, None] + 8 * np.arange(2)[None, :, None]).reshape(32, 8)
    perm = (p1[:, None, :] + 256 * np.arange(4)[None, :, None]).reshape(-1, 8)
    perm = perm[:, [0, 2, 4, 6, 1, 3, 5, 7]].ravel()
    scale_perm = (np.arange(8)[:, None] + 8 * np.arange(8)[None, :]).ravel()
    scale_perm_single = (2 * np.arange(4)[:, None] + np.array([0, 1, 8, 9, 16, 17, 24, 25])[None, :]).ravel()
    return perm, scale_perm, scale_perm_single


def pack_ints(w: np.ndarray, s: np.ndarray, group_size: int):
    """w uint [K, N] in 0..15 (already offset by 8), s fp16 [G, N] natural -> (B int32 [K/16, 2N], s_marlin [G, N])  (:151-169)."""
    K, N = w.shape
    perm, sp, sps = perms()
    if group_size != K:
        sm = s.reshape(-1, 64)[:, sp].reshape(-1, N)
    else:
        sm = s.reshape(-1, 32)[:, sps].reshape(-1, N)
    t = w.reshape(K // 16, 16, N // 16, 16).transpose(0, 2, 1, 3).reshape(K // 16, N * 16)
    t = t.reshape(-1, perm.size)[:, perm].reshape(t.shape).astype(np.uint32)
    q = np.zeros((t.shape[0], t.shape[1] // 8), dtype=np.uint32)
    for i in range(8):
        q |= t[:, i::8] << np.uint32(4 * i)
    return q.view(np.int32), np.ascontiguousarray(sm)


def unpack_ints(B: np.ndarray, s_marlin: np.ndarray, group_size: int):
    """Inverse of pack_ints: -> (w uint8 [K, N], s fp16 [G, N] natural)."""
    K, N = B.shape[0] * 16, B.shape[1] // 2
    perm, sp, sps = perms()
    t = ((B.view(np.uint32)[:, :, None] >> (4 * np.arange(8, dtype=np.uint32))[None, None, :]) & 15).reshape(K // 16, N * 16)
    inv = np.argsort(perm)
    t = t.reshape(-1, perm.size)[:, inv].reshape(K // 16, N // 16, 16, 16).transpose(0, 2, 1, 3).reshape(K, N)
    if group_size != K:
        s = s_marlin.reshape(-1, 64)[:, np.argsort(sp)].reshape(-1, N)
    else:
        s = s_marlin.reshape(-1, 32)[:, np.argsort(sps)].reshape(-1, N)
    return t.astype(np.uint8), np.ascontiguousarray(s)


def to_gptq(B: np.ndarray, s_marlin: np.ndarray, group_size: int):
    """Marlin (symmetric int4, implicit zero-point 8) -> GPTQ v1 tensors: qweight [K/8, N], qzeros [G, N/8] (fields 8 - 1 = 7),
    scales [G, N]."""
    w, s = unpack_ints(B, s_marlin, group_size)
    K, N = w.shape
    qweight = np.zeros((K // 8, N), dtype=np.uint32)
    for j in range(8):
        qweight |= w[j::8].astype(np.uint32) << np.uint32(4 * j)
    qzeros = np.full((s.shape[0], N // 8), 0x77777777, dtype=np.uint32)
    return qweight.view(np.int32), qzeros.view(np.int32), s
