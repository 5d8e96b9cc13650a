"""Fused callers of the hot path: q/k/v as one layer, gate/up as one layer with a SiLU*mul epilogue.

Both are the reference's own constructions:
* fused attention concatenates the packed q/k/v tensors along out_features into ONE QuantLinear
  (auto_gptq/nn_modules/fused_llama_attn.py:171-203) -- nothing new is needed in the kernels for that, a fused layer is
  an ordinary layer with N = N_q + N_k + N_v;
* the fused MLP computes  silu(x @ W_gate) * (x @ W_up)  in one kernel (fused_llama_mlp.py:157-242); here that is
  ``QuantLinear(..., epilogue="silu_mul")`` over the [gate | up] concatenation: for M <= 8 the matrix-core GEMV walks
  both halves in the same workgroup and writes the product (one launch, no [M, 2N] round trip), larger M stage
  y = [gate | up] in the workspace and run an elementwise pass.
"""
from __future__ import annotations

from typing import Sequence

import torch
import torch.nn as nn

from .qlinear_mi355x import QuantLinear


def fuse_quant_linears(layers: Sequence[QuantLinear], epilogue: str = "none") -> QuantLinear:
    """Concatenate quantized linears that read the same input along out_features.

    All layers must agree on in_features, bits, group_size, scales dtype and g_idx.  (With act-order every projection has
    its own g_idx; the reference's cuda backend then carries a 3*K-long g_idx, qlinear_cuda.py:300-312, and its exllama
    backend refuses, fused_llama_attn.py:176-183.  This backend refuses too.)
    """
    if len(layers) < 2:
        raise ValueError("need at least two layers to fuse")
    a = layers[0]
    for l in layers[1:]:
        if (l.infeatures, l.bits, l.group_size) != (a.infeatures, a.bits, a.group_size):
            raise ValueError("fused layers must share in_features, bits and group_size")
        if l.scales.dtype != a.scales.dtype:
            raise ValueError("fused layers must share the scales dtype")
        if (l.bias is None) != (a.bias is None):
            raise ValueError("either every fused layer has a bias or none has")
        if not torch.equal(l.g_idx.cpu(), a.g_idx.cpu()):
            raise ValueError("fused layers must share g_idx (act-order projections with different g_idx cannot be fused)")
        if l.zero_mode != a.zero_mode:
            raise ValueError("fused layers must share zero_mode")
    if epilogue == "silu_mul" and (len(layers) != 2 or layers[0].outfeatures != layers[1].outfeatures):
        raise ValueError("epilogue='silu_mul' fuses exactly two layers (gate, up) of equal width")
    n_total = sum(l.outfeatures for l in layers)
    f = QuantLinear(a.bits, a.group_size, a.infeatures, n_total, a.bias is not None, weight_dtype=a.scales.dtype,
                    zero_mode=a.zero_mode, epilogue=epilogue)
    f.qweight = torch.cat([l.qweight for l in layers], dim=1).contiguous()
    f.qzeros = torch.cat([l.qzeros for l in layers], dim=1).contiguous()
    f.scales = torch.cat([l.scales for l in layers], dim=1).contiguous()
    f.g_idx = a.g_idx.clone()
    if a.bias is not None:
        f.bias = torch.cat([l.bias for l in layers], dim=0).contiguous()
    return f


def fuse_qkv(q: QuantLinear, k: QuantLinear, v: QuantLinear) -> QuantLinear:
    """One layer computing [q | k | v] (split the output with ``torch.split(y, (Nq, Nk, Nv), dim=-1)``)."""
    return fuse_quant_linears([q, k, v])


def fuse_gate_up(gate: QuantLinear, up: QuantLinear) -> QuantLinear:
    """One layer computing silu(gate(x)) * up(x)."""
    return fuse_quant_linears([gate, up], epilogue="silu_mul")


class _FusedQKVState:
    """y = [q | k | v](x) computed once per attention call by the q_proj stand-in and handed to the k/v stand-ins."""

    def __init__(self, fused: QuantLinear, splits):
        self.fused, self.splits = fused, splits
        self.src = None
        self.parts = None


class _QKVPart(nn.Module):
    """Stand-in for q_proj / k_proj / v_proj of an attention module whose forward calls them in that order on the same
    hidden_states (transformers' LlamaAttention.forward does): index 0 runs the fused layer, 1 and 2 return their slices.
    Keeps the host model's attention code untouched, where the reference replaces the whole attention module
    (FusedLlamaAttentionForQuantizedModel, auto_gptq/nn_modules/fused_llama_attn.py:18-135)."""

    def __init__(self, state: _FusedQKVState, index: int, owner: bool):
        super().__init__()
        self.index = index
        self._state = [state]                       # in a list: not registered as a submodule three times
        if owner:
            self.fused = state.fused                # registered once, so .to() / state_dict() see it

    def forward(self, x):
        st = self._state[0]
        if self.index == 0:
            st.src = x
            st.parts = torch.split(st.fused(x), st.splits, dim=-1)
        elif st.src is not x:
            raise RuntimeError("fused q/k/v: k_proj / v_proj called on a different tensor than q_proj")
        return st.parts[self.index]


class FusedGateUpMLP(nn.Module):
    """down(silu(gate(x)) * up(x)) as two launches: the [gate | up] layer with the SiLU*mul epilogue, then down
    (the role of FusedLlamaMLPForQuantizedModel, auto_gptq/nn_modules/fused_llama_mlp.py:131-306)."""

    def __init__(self, gate_up: QuantLinear, down: nn.Module):
        super().__init__()
        self.gate_up = gate_up
        self.down_proj = down

    def forward(self, x):
        return self.down_proj(self.gate_up(x))


def inject_fused_llama(model: nn.Module, fuse_attention: bool = True, fuse_mlp: bool = True) -> int:
    """Fuse q/k/v and gate/up of every Llama-style decoder block whose projections are mi355x QuantLinears with a common g_idx
    (the reference's inject_to_model entry points, fused_llama_attn.py:137-231 / fused_llama_mlp.py:290-306; like its exllama
    backend it leaves blocks with per-projection act-order g_idx unfused).  Returns the number of fused modules."""
    n = 0
    for mod in list(model.modules()):
        if fuse_attention and all(isinstance(getattr(mod, a, None), QuantLinear) for a in ("q_proj", "k_proj", "v_proj")):
            q, k, v = mod.q_proj, mod.k_proj, mod.v_proj
            try:
                f = fuse_qkv(q, k, v)
            except ValueError:
                f = None
            if f is not None:
                f = f.to(q.qweight.device)
                st = _FusedQKVState(f, (q.outfeatures, k.outfeatures, v.outfeatures))
                mod.q_proj, mod.k_proj, mod.v_proj = _QKVPart(st, 0, True), _QKVPart(st, 1, False), _QKVPart(st, 2, False)
                n += 1
        if fuse_mlp and all(isinstance(getattr(mod, a, None), QuantLinear) for a in ("gate_proj", "up_proj")) and hasattr(mod, "down_proj") \
                and getattr(getattr(mod, "act_fn", None), "__class__", type(None)).__name__ in ("SiLU", "SiLUActivation"):
            try:
                gu = fuse_gate_up(mod.gate_proj, mod.up_proj)
            except ValueError:
                gu = None
            if gu is not None:
                fm = FusedGateUpMLP(gu.to(mod.gate_proj.qweight.device), mod.down_proj)
                mod.forward = fm.forward
                mod.fused_mlp = fm
                del mod.gate_proj, mod.up_proj
                n += 1
    return n


__all__ = ["fuse_quant_linears", "fuse_qkv", "fuse_gate_up", "inject_fused_llama", "FusedGateUpMLP"]
