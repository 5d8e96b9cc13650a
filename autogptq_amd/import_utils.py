"""Backend probe + selector, mirroring auto_gptq/utils/import_utils.py:8-112.

In the reference, ``dynamically_import_QuantLinear`` is the single registration point every caller
goes through (make_quant, pack_model, fused injectors, tests).  This mirror keeps its signature and
always answers with the mi355x backend; INTEGRATION.md shows the 6-line branch a maintainer adds to
the reference's own selector to route to it.
"""
from logging import getLogger
from typing import Optional

logger = getLogger(__name__)

try:
    from . import _lib

    _lib.load()
    MI355X_KERNELS_AVAILABLE = True
    MI355X_IMPORT_EXCEPTION = None
except Exception as e:  # library not built yet: keep the probe importable, fail at use
    MI355X_KERNELS_AVAILABLE = False
    MI355X_IMPORT_EXCEPTION = e


def dynamically_import_QuantLinear(
    use_triton: bool = False,
    desc_act: bool = False,
    group_size: int = 128,
    bits: int = 4,
    disable_exllama: Optional[bool] = None,
    disable_exllamav2: bool = False,
    use_qigen: bool = False,
    use_marlin: bool = False,
    use_tritonv2: bool = False,
):
    if use_triton or use_tritonv2 or use_qigen or use_marlin:
        raise ValueError("autogptq_amd only provides the mi355x QuantLinear backend "
                         "(use_triton/use_tritonv2/use_qigen/use_marlin must be False).")
    if bits not in (2, 3, 4, 8):
        raise NotImplementedError("Only 2,3,4,8 bits are supported.")
    if not MI355X_KERNELS_AVAILABLE:
        raise ValueError(f"mi355x kernels are not available: {MI355X_IMPORT_EXCEPTION}. "
                         "Build them with `python -c 'import __graft_entry__ as g; g.build()'`.")
    from .qlinear_mi355x import QuantLinear

    return QuantLinear
