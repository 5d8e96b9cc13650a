"""ctypes binding of libgptq_mi355x.so (the C ABI in include/gptq_mi355x.h).

There is deliberately no fallback: if the shared library is missing or an entry point is absent
this module raises at import/use time -- the product path never silently degrades to PyTorch.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_int, c_int32, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPTQ_MI355X_LIB", os.path.join(_HERE, "libgptq_mi355x.so"))

ABI_VERSION = 7

GPTQ_F16, GPTQ_BF16, GPTQ_F32 = 0, 1, 2
ZERO_WRAP, ZERO_NOWRAP = 0, 1
EPI_NONE, EPI_SILU_MUL = 0, 1

DTYPE_ENUM = {torch.float16: GPTQ_F16, torch.bfloat16: GPTQ_BF16, torch.float32: GPTQ_F32}

# every symbol include/gptq_mi355x.h declares (tests check the .so exports all of them)
EXPORTS = (
    "gptq_abi_version", "gptq_last_error", "gptq_status_string", "gptq_workspace_bytes", "gptq_workspace_bytes_ex",
    "gptq_forward", "gptq_forward_ex", "gptq_gemv", "gptq_gemm", "gptq_dequant",
    "gptq_unpack_weights", "gptq_unpack_zeros", "gptq_pack_weights", "gptq_pack_zeros",
    "gptq_make_sequential", "gptq_resequence_qweight", "gptq_permute_columns", "gptq_prepack_decode", "gptq_prepack_decode_bytes", "gptq_unprepack_decode",
    "gptq_awq_unpack", "gptq_awq_repack", "gptq_describe_plan",
    "gptq_init", "gptq_workspace_bytes_max", "gptq_validate_g_idx",
    "gptq_forward_multi", "gptq_workspace_bytes_multi", "gptq_forward_multi_ex", "gptq_workspace_bytes_multi_ex",
    "gptq_peer_scatter", "gptq_peer_collect", "gptq_peer_gather", "gptq_forward_scatter", "gptq_forward_gather", "gptq_peer_publish",
    "gptq_mlp_forward", "gptq_mlp_forward_ex", "gptq_workspace_bytes_mlp", "gptq_workspace_bytes_mlp_ex", "gptq_describe_mlp_plan",
)
WS_HEADER_BYTES = 65536
STRIP_COLS = 16          # GPTQ_STRIP_COLS: columns per strip of the decode copy (gptq_prepack_decode)


class GptqLayer(Structure):
    _fields_ = [
        ("qweight", c_void_p), ("qzeros", c_void_p), ("scales", c_void_p), ("g_idx", c_void_p), ("bias", c_void_p),
        ("K", c_int32), ("N", c_int32), ("bits", c_int32), ("group_size", c_int32),
        ("dtype", c_int32), ("zero_mode", c_int32),
        ("qweight_seq", c_void_p), ("perm", c_void_p),
        ("epilogue", c_int32), ("tiled_cols", c_int32),
        ("qweight_tiled", c_void_p), ("qconst_tiled", c_void_p),
    ]


class GptqTuning(Structure):
    _fields_ = [("lanes_n", c_int32), ("waves", c_int32), ("ksplit", c_int32), ("path", c_int32),
                ("reserved", c_int32 * 4)]


PEER_MAX = 8


class GptqPeerGroup(Structure):
    """gptq_peer_group_t: the mapped exchange buffers / flags of all ranks, in rank order (include/gptq_mi355x.h)."""
    _fields_ = [("xbuf", (c_void_p * PEER_MAX) * 2), ("flags", c_void_p * PEER_MAX), ("state", c_void_p),
                ("world", c_int32), ("rank", c_int32), ("rows_max", c_int32), ("N", c_int32)]


class GptqError(RuntimeError):
    """Raised for any non-zero status of the C ABI (the reference surfaces native failures as
    RuntimeError through TORCH_CHECK, e.g. autogptq_extension/exllama/exllama_ext.cpp:25-43)."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


class _Lab:
    """Names of the LAB switches carried in GptqTuning.reserved[] (include/gptq_mi355x_lab.h is the one definition: parsed here, prefix GPTQ_LAB_ dropped) --
    ``t.reserved[LAB.GEMM_VARIANT] = LAB.VARIANT_WIDE_SK_ON`` instead of ``t.reserved[3] = 48``.  Not part of the drop-in boundary."""

    def __init__(self):
        path = os.path.join(os.path.dirname(_HERE), "include", "gptq_mi355x_lab.h")
        try:
            text = open(path).read()
        except OSError:
            text = ""
        for m in __import__("re").finditer(r"^#define\s+GPTQ_LAB_(\w+)\s+(\d+)", text, flags=8):
            setattr(self, m.group(1), int(m.group(2)))


LAB = _Lab()

_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            f"or make -C autogptq_amd/csrc). autogptq_amd has no PyTorch/CPU fallback for the quantized matmul.")
    lib = ctypes.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export {missing}")
    lib.gptq_abi_version.restype = c_int
    lib.gptq_last_error.restype = c_char_p
    lib.gptq_status_string.restype = c_char_p
    lib.gptq_status_string.argtypes = [c_int]
    lib.gptq_workspace_bytes.restype = c_size_t
    lib.gptq_workspace_bytes.argtypes = [POINTER(GptqLayer), c_int]
    lib.gptq_workspace_bytes_ex.restype = c_size_t
    lib.gptq_workspace_bytes_ex.argtypes = [POINTER(GptqLayer), c_int, POINTER(GptqTuning)]
    lib.gptq_workspace_bytes_max.restype = c_size_t
    lib.gptq_workspace_bytes_max.argtypes = [POINTER(GptqLayer), c_int]
    lib.gptq_init.argtypes = []
    lib.gptq_workspace_bytes_multi.restype = c_size_t
    lib.gptq_workspace_bytes_multi.argtypes = [POINTER(POINTER(GptqLayer)), c_int, c_int]
    lib.gptq_forward_multi.argtypes = [POINTER(POINTER(GptqLayer)), c_int, c_void_p, POINTER(c_void_p), c_int, c_void_p, c_size_t, c_void_p]
    lib.gptq_workspace_bytes_multi_ex.restype = c_size_t
    lib.gptq_workspace_bytes_multi_ex.argtypes = [POINTER(POINTER(GptqLayer)), c_int, c_int, POINTER(GptqTuning)]
    lib.gptq_forward_multi_ex.argtypes = lib.gptq_forward_multi.argtypes + [POINTER(GptqTuning)]
    lib.gptq_validate_g_idx.argtypes = [c_void_p, c_int, c_int]
    fw = [POINTER(GptqLayer), c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]
    lib.gptq_forward.argtypes = fw
    for name in ("gptq_forward_ex", "gptq_gemv", "gptq_gemm"):
        getattr(lib, name).argtypes = fw + [POINTER(GptqTuning)]
    lib.gptq_dequant.argtypes = [POINTER(GptqLayer), c_void_p, c_void_p]
    lib.gptq_unpack_weights.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.gptq_unpack_zeros.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.gptq_pack_weights.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_void_p, c_void_p, c_void_p]
    lib.gptq_pack_zeros.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.gptq_make_sequential.argtypes = [c_void_p, c_int, c_int, c_void_p, POINTER(c_int)]
    lib.gptq_resequence_qweight.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.gptq_permute_columns.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.gptq_prepack_decode_bytes.argtypes = [POINTER(GptqLayer), POINTER(c_size_t), POINTER(c_size_t)]
    lib.gptq_prepack_decode.argtypes = [POINTER(GptqLayer), c_void_p, c_void_p, c_void_p]
    lib.gptq_unprepack_decode.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.gptq_describe_plan.argtypes = [POINTER(GptqLayer), c_int, POINTER(GptqTuning), c_char_p, c_size_t]
    lib.gptq_awq_unpack.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    lib.gptq_awq_repack.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
    LP = POINTER(GptqLayer)
    lib.gptq_workspace_bytes_mlp.restype = c_size_t
    lib.gptq_workspace_bytes_mlp.argtypes = [LP, LP, LP, c_int]
    lib.gptq_workspace_bytes_mlp_ex.restype = c_size_t
    lib.gptq_workspace_bytes_mlp_ex.argtypes = [LP, LP, LP, c_int, POINTER(GptqTuning)]
    lib.gptq_mlp_forward.argtypes = [LP, LP, LP, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]
    lib.gptq_mlp_forward_ex.argtypes = lib.gptq_mlp_forward.argtypes + [POINTER(GptqTuning)]
    lib.gptq_describe_mlp_plan.argtypes = [LP, LP, LP, c_int, POINTER(GptqTuning), c_char_p, c_size_t]
    lib.gptq_peer_scatter.argtypes = [POINTER(GptqPeerGroup), c_void_p, c_int, c_int, c_int, c_void_p]
    lib.gptq_peer_collect.argtypes = [POINTER(GptqPeerGroup), c_void_p, c_int, c_int, ctypes.c_uint32, c_void_p]
    lib.gptq_peer_gather.argtypes = [POINTER(GptqPeerGroup), c_void_p, c_void_p, c_int, c_int, c_int, ctypes.c_uint32, c_void_p]
    lib.gptq_peer_publish.argtypes = [POINTER(GptqPeerGroup), c_void_p]
    lib.gptq_forward_scatter.argtypes = [LP, c_void_p, c_int, POINTER(GptqPeerGroup), c_void_p, c_size_t, c_void_p]
    lib.gptq_forward_gather.argtypes = [LP, c_void_p, c_void_p, c_int, POINTER(GptqPeerGroup), ctypes.c_uint32, c_void_p, c_size_t, c_void_p]
    for name in EXPORTS:
        if name not in ("gptq_last_error", "gptq_status_string", "gptq_workspace_bytes", "gptq_workspace_bytes_ex",
                        "gptq_workspace_bytes_max", "gptq_workspace_bytes_multi", "gptq_workspace_bytes_multi_ex",
                        "gptq_workspace_bytes_mlp", "gptq_workspace_bytes_mlp_ex"):
            getattr(lib, name).restype = c_int
    got = lib.gptq_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{LIB_PATH}: ABI version {got}, expected {ABI_VERSION}")
    _lib = lib
    _bind_fastcall(lib)
    return lib


fast = None      # autogptq_amd/_fastcall.so (METH_FASTCALL trampoline), or None: the same entry points through ctypes


def _bind_fastcall(lib) -> None:
    """Hand the addresses of the two per-token entry points to the C trampoline, if it was built (python -c 'import
    __graft_entry__ as g; g.build()').  It calls the SAME functions of the SAME loaded library; only the argument marshalling differs."""
    global fast
    try:
        from . import _fastcall
    except ImportError:
        fast = None
        _bind_fastfwd(lib)
        return
    try:
        _fastcall.bind(ctypes.cast(lib.gptq_forward_ex, c_void_p).value, ctypes.cast(lib.gptq_forward_multi_ex, c_void_p).value,
                       ctypes.cast(lib.gptq_mlp_forward, c_void_p).value)
    except TypeError:          # a _fastcall.so built from an older revision (two-argument bind): use ctypes rather than fail the load
        fast = None
        return
    fast = _fastcall
    _bind_fastfwd(lib)


fwd = None       # autogptq_amd/_fastfwd.so (the eager per-call path in C++ on ATen), or None: the Python path does it all


def _bind_fastfwd(lib) -> None:
    """The C++ fast path of QuantLinear.forward / forward_multi (cext/fastfwd.cpp), if it was built for this torch: the same two C-ABI functions of the
    same loaded library behind one METH_FASTCALL call that also allocates the output and reads the current stream.  Absent or unloadable (another torch
    version): the Python path is complete without it."""
    global fwd
    fwd = None
    if os.environ.get("GPTQ_MI355X_NO_FASTFWD"):
        return
    try:
        import torch  # noqa: F401  (libtorch_python must be loaded first)
        from . import _fastfwd
        _fastfwd.bind(ctypes.cast(lib.gptq_forward_ex, c_void_p).value, ctypes.cast(lib.gptq_forward_multi_ex, c_void_p).value)
    except Exception:
        return
    fwd = _fastfwd


def check(status: int) -> None:
    if status != 0:
        lib = load()
        msg = lib.gptq_last_error().decode("utf-8", "replace")
        kind = lib.gptq_status_string(status).decode()
        raise GptqError(status, f"gptq_mi355x: {kind}: {msg}")


def describe_plan(layer: "GptqLayer", M: int, tuning: "GptqTuning | None" = None) -> dict:
    """Kernel and launch geometry gptq_forward_ex would pick (host-only query, see gptq_describe_plan in the header)."""
    lib = load()
    buf = ctypes.create_string_buffer(512)
    check(lib.gptq_describe_plan(ctypes.byref(layer), M, ctypes.byref(tuning) if tuning is not None else None, buf, len(buf)))
    out = {}
    for kv in buf.value.decode().split():
        k, v = kv.split("=", 1)
        out[k] = int(v) if v.lstrip("-").isdigit() else v
    return out


def describe_mlp_plan(gate: "GptqLayer", up: "GptqLayer", down: "GptqLayer", M: int, tuning: "GptqTuning | None" = None) -> dict:
    """What gptq_mlp_forward[_ex] would run for these three layers (host-only query)."""
    lib = load()
    buf = ctypes.create_string_buffer(512)
    check(lib.gptq_describe_mlp_plan(ctypes.byref(gate), ctypes.byref(up), ctypes.byref(down), M,
                                     ctypes.byref(tuning) if tuning is not None else None, buf, len(buf)))
    out = {}
    for kv in buf.value.decode().split():
        k, v = kv.split("=", 1)
        out[k] = int(v) if v.lstrip("-").isdigit() else v
    return out


_INITED = set()


def ensure_init(device) -> None:
    """gptq_init() once per device (outside stream capture: QuantLinear.post_init calls this)."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx in _INITED:
        return
    with torch.cuda.device(idx):
        check(load().gptq_init())
    _INITED.add(idx)


def ptr(t):
    return None if t is None else t.data_ptr()


def current_stream_handle(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream
