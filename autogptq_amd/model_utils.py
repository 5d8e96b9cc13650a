"""The callers either side of the hot path, mirrored so that a model can be switched to this backend
without the rest of AutoGPTQ:

* ``make_quant``           auto_gptq/modeling/_utils.py:69-147   -- swap nn.Linear / Conv1D modules for QuantLinear
* ``pack_model``           auto_gptq/modeling/_utils.py:257-330  -- run QuantLinear.pack over a dict of quantizer outputs
* ``autogptq_post_init``   auto_gptq/modeling/_utils.py:380-513  -- per-layer post_init + one scratch buffer per device
* ``load_packed_layers``   auto_gptq/modeling/_base.py:1040-1140  -- fill the swapped layers from GPTQ / Marlin / AWQ tensors

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
        # the need is not monotone in M (K splits come and go with the kernel the planner picks): maximum over 1..rows
        b = int(lib.gptq_workspace_bytes_max(ctypes.byref(sub._layer), rows))
        need[sub.qweight.device] = max(need.get(sub.qweight.device, 0), b)
    for dev, b in need.items():
        reserve_workspace(dev, b)
    return model


def load_packed_layers(model: nn.Module, state_dict: Dict[str, torch.Tensor], bits: int, group_size: int, desc_act: bool = False,
                       quant_method: str = "gptq", checkpoint_format: str = "gptq") -> nn.Module:
    """Fill a model skeleton from quantized-layer tensors in any of the three layouts the reference serialises
    (``quant_method`` / ``checkpoint_format`` as in ``quantize_config.json``, auto_gptq/quantization/config.py:24-46):

    * ``gptq`` / ``gptq``   ``<name>.{qweight,qzeros,scales,g_idx[,bias]}``  -- copied as they are
    * ``gptq`` / ``marlin`` ``<name>.{B,s[,bias]}``                          -- ``marlin.marlin_to_gptq`` (index arithmetic)
    * ``awq``  / ``gemm``   ``<name>.{qweight [K, N/8],qzeros,scales[,bias]}`` -- ``awq.repack_awq_to_gptq`` (HIP kernel; the
      layer is then read with the cuda_old zero convention, which maps the stored (z - 1) & 15 back to z)

    The named linears are swapped for this backend's QuantLinear (``make_quant``); every other entry of ``state_dict`` is
    loaded non-strictly into the rest of the model.  This is the part of ``from_quantized`` (auto_gptq/modeling/_base.py:
    1040-1140) that touches the hot path's tensors; file discovery, device maps and the model zoo stay with the caller."""
    from . import awq as _awq
    from . import marlin as _marlin
    if (quant_method, checkpoint_format) not in (("gptq", "gptq"), ("gptq", "marlin"), ("awq", "gemm")):
        raise ValueError(f"The checkpoint format used is {checkpoint_format}, and the quantization method is {quant_method}. "
                         "This is not supported.")
    if quant_method == "awq" or checkpoint_format == "marlin":
        if bits != 4:
            raise ValueError("Marlin and AWQ checkpoints are 4-bit")
    probe = "B" if checkpoint_format == "marlin" else "qweight"
    names = sorted(k[: -len(probe) - 1] for k in state_dict if k.endswith("." + probe))
    linears = find_layers(model)
    missing = [n for n in names if n not in linears]
    if missing:
        raise KeyError(f"quantized tensors for modules the model does not have: {missing[:4]}")
    make_quant(model, names, bits, group_size, desc_act=desc_act)
    qlayers = find_layers(model, [QuantLinear])
    used = set()
    for name in names:
        q = qlayers[name]
        gs = q.group_size
        take = lambda suffix: state_dict[f"{name}.{suffix}"]                       # noqa: E731
        if checkpoint_format == "marlin":
            qweight, qzeros, scales = _marlin.marlin_to_gptq(take("B"), take("s"), gs)
            used.update({f"{name}.B", f"{name}.s"})
        elif quant_method == "awq":
            qweight, qzeros = _awq.repack_awq_to_gptq(take("qweight"), take("qzeros"), gs)
            scales = take("scales")
            q.zero_mode = "wrap"
            used.update({f"{name}.qweight", f"{name}.qzeros", f"{name}.scales"})
        else:
            qweight, qzeros, scales = take("qweight"), take("qzeros"), take("scales")
            used.update({f"{name}.qweight", f"{name}.qzeros", f"{name}.scales"})
            if f"{name}.g_idx" in state_dict:
                q.g_idx = state_dict[f"{name}.g_idx"].to(torch.int32).to(q.g_idx.device)
                used.add(f"{name}.g_idx")
        for attr, t in (("qweight", qweight), ("qzeros", qzeros), ("scales", scales)):
            cur = getattr(q, attr)
            if tuple(t.shape) != tuple(cur.shape):
                raise ValueError(f"{name}.{attr}: checkpoint has shape {tuple(t.shape)}, the layer expects {tuple(cur.shape)}")
            setattr(q, attr, t.to(device=cur.device, dtype=cur.dtype).contiguous())
        if f"{name}.bias" in state_dict and q.bias is not None:
            q.bias = state_dict[f"{name}.bias"].to(device=q.bias.device, dtype=q.bias.dtype)
            used.add(f"{name}.bias")
        q._invalidate()
    rest = {k: v for k, v in state_dict.items() if k not in used and not k.endswith(".workspace")}
    if rest:
        model.load_state_dict(rest, strict=False)
    return model


__all__ = ["make_quant", "pack_model", "autogptq_post_init", "find_layers", "load_packed_layers"]
