"""ctypes binding of oracle/libgptq_oracle.so (C restatement) -- TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libgptq_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gptq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t)) if a is not None else None


def unpack_weights(qweight: np.ndarray, bits: int) -> np.ndarray:
    q = np.ascontiguousarray(qweight, dtype=np.int32)
    K, N = q.shape[0] * 32 // bits, q.shape[1]
    out = np.empty((K, N), np.uint16)
    rc = lib().gptq_oracle_unpack_weights(_p(q, ctypes.c_int32), K, N, bits, _p(out, ctypes.c_uint16))
    assert rc == 0
    return out


def unpack_zeros(qzeros: np.ndarray, bits: int, nowrap: bool) -> np.ndarray:
    q = np.ascontiguousarray(qzeros, dtype=np.int32)
    G, N = q.shape[0], q.shape[1] * 32 // bits
    out = np.empty((G, N), np.int32)
    rc = lib().gptq_oracle_unpack_zeros(_p(q, ctypes.c_int32), G, N, bits, int(nowrap), _p(out, ctypes.c_int32))
    assert rc == 0
    return out


def forward_f64(x, qweight, qzeros, scales, g_idx, bias, bits, group_size, nowrap) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.ascontiguousarray(qweight, dtype=np.int32)
    qz = np.ascontiguousarray(qzeros, dtype=np.int32)
    s = np.ascontiguousarray(scales, dtype=np.float32)
    g = None if g_idx is None else np.ascontiguousarray(g_idx, dtype=np.int32)
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    M, K = x.shape
    N = q.shape[1]
    y = np.empty((M, N), np.float64)
    rc = lib().gptq_oracle_forward_f64(_p(x, ctypes.c_float), _p(q, ctypes.c_int32), _p(qz, ctypes.c_int32),
                                       _p(s, ctypes.c_float), _p(g, ctypes.c_int32), _p(b, ctypes.c_float),
                                       M, K, N, bits, group_size, int(nowrap), _p(y, ctypes.c_double))
    assert rc == 0
    return y


def decode_copy_weights(qweight: np.ndarray, bits: int) -> np.ndarray:
    """uint32 [N/16, chunks, 4, 16, WPL] (include/gptq_mi355x.h: qweight_tiled)."""
    q = np.ascontiguousarray(qweight, dtype=np.int32)
    K, N = q.shape[0] * 32 // bits, q.shape[1]
    kpl, wpl = (16 if bits == 8 else 32), (3 if bits == 3 else (2 if bits == 2 else 4))
    out = np.empty((N // 16, -(-K // (4 * kpl)), 4, 16, wpl), np.uint32)
    rc = lib().gptq_oracle_decode_copy_weights(_p(q, ctypes.c_int32), K, N, bits, _p(out, ctypes.c_uint32))
    assert rc == 0
    return out


def decode_copy_consts(qzeros: np.ndarray, scale_bits: np.ndarray, bits: int, nowrap: bool) -> np.ndarray:
    """uint8 [N/16, G, REC] (include/gptq_mi355x.h: qconst_tiled); scale_bits = the 16-bit scales viewed as uint16 [G, N]."""
    q = np.ascontiguousarray(qzeros, dtype=np.int32)
    sb = np.ascontiguousarray(scale_bits, dtype=np.uint16)
    G, N = sb.shape
    out = np.empty((N // 16, G, 64 if bits == 8 else 48), np.uint8)
    rc = lib().gptq_oracle_decode_copy_consts(_p(q, ctypes.c_int32), _p(sb, ctypes.c_uint16), G, N, bits, int(nowrap), _p(out, ctypes.c_uint8))
    assert rc == 0
    return out
