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


def autogptq_post_init(model: nn.Module, use_act_order: bool = False, max_input_length: Optional[int] = None,
                       release_checkpoint_layout: Optional[bool] = None, decode_copy: Optional[bool] = None) -> nn.Module:
    """post_init every mi355x layer and size the per-device scratch once (so forward never allocates; needed before hipGraph
    capture).  ``max_input_length`` bounds the rows M the scratch is sized for (default 2048, the reference's exllama default).
    Memory (the model-level switches for what post_init keeps next to the checkpoint tensors): ``decode_copy=False`` builds no decode copy (1x the packed
    bytes, the round-1..3 kernels); ``release_checkpoint_layout=True`` keeps the copy and moves ``qweight`` of plain layers to pinned host memory (1x on
    the device again; ``state_dict()`` unchanged; row counts whose kernel reads packed rows rebuild them per call into one shared scratch).  Default: both
    layouts on the device (2x; act-order layers 3x with their re-sequenced rows) -- 288 GB of HBM is what makes that the default."""
    rows = max_input_length or 2048
    need: Dict[torch.device, int] = {}
    for _, sub in model.named_modules():
        if getattr(sub, "QUANT_TYPE", None) != QuantLinear.QUANT_TYPE:
            continue
        if sub.qweight.device.type != "cuda":
            continue
        dev = sub.qweight.device                      # read BEFORE post_init: release_checkpoint_layout moves qweight to pinned host memory, and the scratch belongs to the layer's GPU
        sub.post_init(tiled=decode_copy, release_checkpoint_layout=release_checkpoint_layout)
        lib = _lib.load()
        # the need is not monotone in M (K splits come and go with the kernel the planner picks): maximum over 1..rows
        parts = getattr(sub, "_parts", None)
        if parts is not None:
            # fused-QKV g_idx (len n * K): the module runs its n column blocks through forward_multi -- the blocks' own needs and the one-launch need of the group
            b = max(int(lib.gptq_workspace_bytes_max(ctypes.byref(p._layer), rows)) for p in parts)
            arr = (ctypes.POINTER(_lib.GptqLayer) * len(parts))(*[ctypes.pointer(p._layer) for p in parts])
            for m in range(1, rows + 1):                  # host arithmetic only (a few microseconds per query)
                b = max(b, int(lib.gptq_workspace_bytes_multi(arr, len(parts), m)))
        else:
            b = int(lib.gptq_workspace_bytes_max(ctypes.byref(sub._layer), rows))
        need[dev] = max(need.get(dev, 0), b)
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


class DecodeStepGraph:
    """One decode step of a causal LM with a STATIC key/value cache as ONE hipGraph (capture_decode_step).  ``graph(input_ids[, cache_position])`` copies the
    small inputs into the captured buffers, replays and returns the captured logits tensor (overwritten by the next replay: clone it to keep it)."""

    def __init__(self, graph, ids, pos, logits, cache):
        self.graph, self.input_ids, self.cache_position, self.logits, self.past_key_values = graph, ids, pos, logits, cache

    def __call__(self, input_ids: torch.Tensor, cache_position: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.input_ids.copy_(input_ids.reshape(self.input_ids.shape))
        if self.cache_position is not None:
            if cache_position is None:
                raise ValueError("this model takes cache_position: pass the position of the token")
            self.cache_position.copy_(cache_position.reshape(self.cache_position.shape))
        self.graph.replay()
        return self.logits


def capture_decode_step(model: nn.Module, past_key_values, batch_size: int = 1, warmup: int = 2) -> DecodeStepGraph:
    """Capture ``model(input_ids[B, 1], past_key_values=<static cache>)`` -- every QuantLinear launch of a decode step (128 per token on Llama-7B with the
    grouped q|k|v / gate|up calls) plus the attention, norms and the lm_head around them -- into one hipGraph.

    Why: the reference's ``generate()`` is eager (auto_gptq/modeling/_base.py:415-418) and a decode launch of this backend runs for 4.5 - 11 us, which is
    what one eager Python call costs on the host; replayed from a graph the same launches run back to back (bench.py's headline is measured that way).
    Needs a cache of fixed shape (``transformers.StaticCache(config, max_cache_len)``): the graph bakes addresses in.  Call ``autogptq_post_init`` first (the
    workspace is sized before capture; nothing is allocated by forward inside it).

    Where the token lands: transformers 5 keeps the write position in the cache itself (``StaticLayer.cumulative_length``, a device tensor every update
    advances in place -- the captured step does that too, so successive replays walk through the cache); the ``warmup`` eager steps run here advance it as
    well and are undone (they need ``warmup`` free slots behind the current length; what they wrote there is beyond the restored length and overwritten
    by the real steps).  Models whose forward still takes ``cache_position`` (transformers 4, the reference's pin) get it as a second captured input;
    warm-up and capture then run at the cache's LAST slot.
    Everything else the model does in a step has to be capturable too: with transformers 5 use ``model.set_attn_implementation("sdpa")`` (its eager-attention
    mask path creates a device scalar from a Python float per call -- a host copy, refused under capture).

    Use::

        cache = StaticCache(model.config, max_cache_len=L)
        logits = model(prompt_ids, past_key_values=cache, use_cache=True).logits              # eager prefill
        step = capture_decode_step(model, cache)
        for _ in range(new_tokens):
            tok = step(tok.view(1, 1))[:, -1].argmax(-1)
    """
    import inspect
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("capture_decode_step needs the model on a ROCm GPU")
    takes_pos = "cache_position" in inspect.signature(model.forward).parameters
    ids = torch.zeros((batch_size, 1), dtype=torch.long, device=dev)
    pos = None
    if takes_pos:
        last = None
        for getter in ("get_max_length", "get_max_cache_shape"):
            if last is None and hasattr(past_key_values, getter):
                try:
                    v = getattr(past_key_values, getter)()
                    last = int(v) - 1 if v is not None and int(v) > 0 else None
                except Exception:
                    last = None
        if last is None and hasattr(past_key_values, "max_cache_len"):
            last = int(past_key_values.max_cache_len) - 1
        if last is None or last < 0:
            raise ValueError("capture_decode_step needs a static key/value cache (fixed max_cache_len)")
        pos = torch.full((1,), last, dtype=torch.long, device=dev)
    counters = [l.cumulative_length for l in getattr(past_key_values, "layers", []) if torch.is_tensor(getattr(l, "cumulative_length", None))]
    saved = [c.clone() for c in counters]

    def step():
        kw = dict(input_ids=ids, past_key_values=past_key_values, use_cache=True, return_dict=True)
        if pos is not None:
            kw["cache_position"] = pos
        return model(**kw).logits

    cur = torch.cuda.current_stream(dev)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(cur)
    with torch.cuda.stream(side), torch.no_grad():
        for _ in range(max(1, warmup)):           # plans, workspace needs and the per-stream scratch are resolved here, outside the capture
            step()
        for c, v in zip(counters, saved):         # the warm-up steps advanced the cache's write position: put it back
            c.copy_(v)
    cur.wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(g):
        logits = step()
    return DecodeStepGraph(g, ids, pos, logits, past_key_values)


__all__ = ["make_quant", "pack_model", "autogptq_post_init", "find_layers", "load_packed_layers", "capture_decode_step", "DecodeStepGraph"]
