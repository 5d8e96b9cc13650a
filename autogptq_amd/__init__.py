"""autogptq_amd -- the AutoGPTQ quantized-linear hot path, MI355X (gfx950) native.

Only what the path needs: the HIP kernels + C ABI (``csrc/`` -> ``libgptq_mi355x.so``), a ctypes
loader (``_lib``), the ``QuantLinear`` backend class (``qlinear_mi355x``), the backend selector
mirror (``import_utils``), the callers either side of the path (``model_utils``, ``fused``), the
checkpoint formats that feed it (``awq``, ``marlin``) and the tensor-parallel wrappers (``tensor_parallel``).
"""
from .import_utils import MI355X_KERNELS_AVAILABLE, dynamically_import_QuantLinear  # noqa: F401
from .qlinear_mi355x import QuantLinear, reserve_workspace  # noqa: F401
from .fused import fuse_gate_up, fuse_qkv, fuse_quant_linears  # noqa: F401
from .model_utils import autogptq_post_init, load_packed_layers, make_quant, pack_model  # noqa: F401

__version__ = "0.1.0"
