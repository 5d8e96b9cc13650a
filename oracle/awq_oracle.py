"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's AWQ -> GPTQ ingest (never imported by the product).

Follows auto_gptq/modeling/_utils.py: awq_reverse_reorder_int_tensor :525-553, unpack_awq :556-621,
pack_from_tensors :624-701.  Pinned by tests/golden/awq_*.npz, which tests/golden/make_golden_awq.py produced by
executing those three reference functions themselves (tests/test_oracle_golden.py checks every array).
"""
import numpy as np

ORDER_MAP = np.array([0, 2, 4, 6, 1, 3, 5, 7])           # _utils.py:533


def reverse_reorder(int_tensor: np.ndarray) -> np.ndarray:
    """_utils.py:525-553: transpose, then columns [order[order]] inside every run of 8."""
    t = np.ascontiguousarray(int_tensor.T)
    n = t.shape[1]
    order = (ORDER_MAP[None, :] + np.arange(0, n, 8)[:, None]).reshape(-1)
    rev = np.arange(n)[order][order]
    return t[:, rev]


def unpack_awq(awq_qweight: np.ndarray, awq_qzeros: np.ndarray, awq_scales: np.ndarray, group_size: int):
    """_utils.py:556-621.  Returns (fp16_weight [N, K] float16, zeros [G, N] int8)."""
    K, NW = awq_qweight.shape
    wf = np.arange(0, 32, 4, dtype=np.uint32)
    qz = awq_qzeros.view(np.uint32)
    zeros = ((qz[:, :, None] >> wf[None, None, :]) & 15).astype(np.int8)             # [G, NW, 8]   (:580-587, no +1)
    zeros = zeros.reshape(-1, NW * 8)                                                # (:589,598)
    qw = np.ascontiguousarray(awq_qweight.T).view(np.uint32)                         # [NW, K]      (:575)
    weight = ((qw[:, None, :] >> wf[None, :, None]) & 15).astype(np.int8)            # [NW, 8, K]   (:591-594)
    weight = weight.reshape(-1, K)                                                   # [N, K] rows = word*8 + nibble (:595-597)
    zeros = reverse_reorder(np.ascontiguousarray(zeros.T))                           # [G, N]       (:600-601)
    weight = reverse_reorder(weight)                                                 # [K, N]       (:602)
    scales = awq_scales.astype(np.float16)
    scale_zeros = (zeros.astype(np.float32) * scales.astype(np.float32)).astype(np.float16)      # int8 * half -> half (:607)
    g_idx = np.arange(K) // group_size
    ws = (weight.astype(np.float32) * scales[g_idx].astype(np.float32)).astype(np.float16)        # (:613)
    qdq = (ws.astype(np.float32) - scale_zeros[g_idx].astype(np.float32)).astype(np.float16)
    return np.ascontiguousarray(qdq.T), zeros


def pack_from_tensors(weight_nk: np.ndarray, zeros: np.ndarray, awq_scales: np.ndarray, group_size: int):
    """_utils.py:624-701 for fp16 inputs: every half op = fp32 op rounded once to half; torch.round = half-to-even."""
    N, K = weight_nk.shape
    s = awq_scales.astype(np.float16).T                                              # [N, G]
    sz = (zeros.T.astype(np.float32) * s.astype(np.float32)).astype(np.float16)      # (:656)
    g = np.arange(K) // group_size
    t = (weight_nk.astype(np.float32) + sz[:, g].astype(np.float32)).astype(np.float16)
    t = (t.astype(np.float32) / s[:, g].astype(np.float32)).astype(np.float16)
    intweight = np.rint(t.astype(np.float32)).astype(np.int32).T.astype(np.uint32)   # [K, N]   (:665-668)
    qweight = np.zeros((K // 8, N), dtype=np.uint32)
    for j in range(8):
        qweight |= intweight[j::8] << np.uint32(4 * j)                               # (:672-677)
    zq = ((zeros.astype(np.int32) - 1) & 15).astype(np.uint32)                       # (:682-683)
    qzeros = np.zeros((zeros.shape[0], N // 8), dtype=np.uint32)
    for j in range(8):
        qzeros |= zq[:, j::8] << np.uint32(4 * j)                                    # (:690-696)
    return qweight.view(np.int32), qzeros.view(np.int32)


def awq_to_gptq(awq_qweight: np.ndarray, awq_qzeros: np.ndarray):
    """The integer composition of the two (what gptq_awq_repack computes)."""
    K, NW = awq_qweight.shape
    pos = np.array([0, 4, 1, 5, 2, 6, 3, 7], dtype=np.uint32) * 4
    w = ((awq_qweight.view(np.uint32)[:, :, None] >> pos[None, None, :]) & 15).reshape(K, NW * 8)
    z = ((awq_qzeros.view(np.uint32)[:, :, None] >> pos[None, None, :]) & 15).reshape(-1, NW * 8)
    qweight = np.zeros((K // 8, NW * 8), dtype=np.uint32)
    for j in range(8):
        qweight |= w[j::8] << np.uint32(4 * j)
    zq = (z.astype(np.int32) - 1) & 15
    qzeros = np.zeros((z.shape[0], NW), dtype=np.uint32)
    for j in range(8):
        qzeros |= zq[:, j::8].astype(np.uint32) << np.uint32(4 * j)
    return qweight.view(np.int32), qzeros.view(np.int32)


def awq_pack(vals: np.ndarray) -> np.ndarray:
    """[R, N] integers in 0..15 -> AWQ words [R, N/8] (AutoAWQ's order: nibble p = column 8c + ORDER_MAP[p])."""
    R, N = vals.shape
    v = vals.reshape(R, N // 8, 8).astype(np.uint32)
    out = np.zeros((R, N // 8), dtype=np.uint32)
    for p in range(8):
        out |= v[:, :, ORDER_MAP[p]] << np.uint32(4 * p)
    return out.view(np.int32)
