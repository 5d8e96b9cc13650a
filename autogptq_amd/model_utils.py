"""The callers either side of the hot path, mirrored so that a model can be switched to this backend
without the rest of AutoGPTQ:

* ``make_quant``           auto_gptq/modeling/_utils.py:69-147   -- swap nn.Linear / Conv1D modules for QuantLinear
* ``pack_model``           auto_gptq/modeling/_utils.py:257-330  -- run QuantLinear.pack over a dict of quantizer outputs
* ``autogptq_post_init``   auto_gptq/modeling/_utils.py:380-513  -- per-layer post_init + one scratch buffer per device

Nothing here touches the kernels; it is wiring with the reference's argument names and defaults.
"""
from __future__ import annotations

import ctypes
from logging import getLogger
from typing import Dict, Iterable, Optional

import torch
import torch.nn as nn

from . import _lib
from .import_utils import dynamically_import_QuantLinear
from .qlinear_mi355x import QuantLinear, reserve_workspace

logger = getLogger(__name__)


def _recurse_setattr(module: nn.Module, name: str, value: nn.Module) -> None:
    if "." not in name:
        setattr(module, name, value)
    else:
        head, rest = name.split(".", 1)
        _recurse_setattr(getattr(module, head), rest, value)


def find_layers(module: nn.Module, layers=None, name: str = "") -> Dict[str, nn.Module]:
    """auto_gptq/modeling/_utils.py:49-60."""
    import transformers
    if not layers:
        layers = [transformers.pytorch_utils.Conv1D, nn.Conv2d, nn.Linear]
    for layer in layers:
        if isinstance(module, layer):
            return {name: module}
    res = {}
    for name1, child in module.named_children():
        res.update(find_layers(child, layers=layers, name=name + "." + name1 if name != "" else name1))
    return res


def make_quant(module: nn.Module, names: Iterable[str], bits: int, group_size: int, name: str = "", use_triton: bool = False,
               use_marlin: bool = False, disable_exllama: Optional[bool] = None, disable_exllamav2: bool = False,
               use_qigen: bool = False, use_cuda_fp16: bool = True, desc_act: bool = False, trainable: bool = False,
               use_tritonv2: bool = False) -> None:
    import transformers
    QL = dynamically_import_QuantLinear(use_triton=use_triton, desc_act=desc_act, group_size=group_size, bits=bits,
                                        use_marlin=use_marlin, disable_exllama=disable_exllama,
                                        disable_exllamav2=disable_exllamav2, use_qigen=use_qigen, use_tritonv2=use_tritonv2)
    if isinstance(module, QL):
        return
    names = set(names)
    for sub_name, sub in list(module.named_modules()):
        if sub_name not in names:
            continue
        dev = next(sub.parameters()).device
        if isinstance(sub, nn.Linear):
            k, n = sub.in_features, sub.out_features
        elif isinstance(sub, nn.Conv2d):
            k, n = sub.in_channels, sub.out_channels
        elif isinstance(sub, transformers.pytorch_utils.Conv1D):
            k, n = sub.weight.shape[0], sub.weight.shape[1]
        else:
            raise TypeError(f"{sub_name}: unsupported module type {type(sub).__name__}")
        new = QL(bits, group_size, k, n, sub.bias is not None, use_cuda_fp16=use_cuda_fp16, trainable=trainable,
                 weight_dtype=sub.weight.dtype)
        new.device = dev
        _recurse_setattr(module, sub_name, new.to(dev))


def pack_model(model: nn.Module, quantizers: dict, bits: int, group_size: int, desc_act: bool = False, **kwargs) -> None:
    """``quantizers[name] = (quantizer, scale, zero, g_idx)`` as produced by GPTQ (auto_gptq/modeling/_base.py:417-420)."""
    layers = {n: layers_n for n, layers_n in find_layers(model).items() if n in quantizers}
    make_quant(model, quantizers, bits, group_size, desc_act=desc_act)
    qlayers = find_layers(model, [QuantLinear])
    for name in qlayers:
        _, scale, zero, g_idx = quantizers[name]
        qlayers[name].pack(layers[name], scale, zero, g_idx)


def autogptq_post_init(model: nn.Module, use_act_order: bool = False, max_input_length: Optional[int] = None) -> nn.Module:
    """post_init every mi355x layer and size the per-device scratch once (so forward never allocates; needed before hipGraph
    capture).  ``max_input_length`` bounds the rows M the scratch is sized for (default 2048, the reference's exllama default)."""
    rows = max_input_length or 2048
    need: Dict[torch.device, int] = {}
    for _, sub in model.named_modules():
        if getattr(sub, "QUANT_TYPE", None) != QuantLinear.QUANT_TYPE:
            continue
        if sub.qweight.device.type != "cuda":
            continue
        sub.post_init()
        lib = _lib.load()
        for m in {1, 8, 64, rows}:
            b = int(lib.gptq_workspace_bytes(ctypes.byref(sub._layer), m))
            need[sub.qweight.device] = max(need.get(sub.qweight.device, 0), b)
    for dev, b in need.items():
        reserve_workspace(dev, b)
    return model


__all__ = ["make_quant", "pack_model", "autogptq_post_init", "find_layers"]
