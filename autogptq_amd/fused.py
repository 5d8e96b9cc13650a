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
    """y_q, y_k, y_v computed by ONE gptq_forward_multi call when the q_proj stand-in runs, handed to the k/v stand-ins."""

    def __init__(self, layers):
        self.layers = list(layers)
        self.src = None
        self.parts = None


class _QKVPart(nn.Module):
    """Stand-in for q_proj / k_proj / v_proj of an attention module whose forward calls them in that order on the same
    hidden_states (transformers' LlamaAttention.forward does): index 0 runs all three projections through
    ``forward_multi`` -- one launch for decode rows on plain layers, three ordinary calls otherwise (act-order, prefill) --
    and 1 / 2 return their results.  The three QuantLinears stay what they were: no concatenated copy of the packed tensors
    (the reference builds one, fused_llama_attn.py:171-203, and therefore cannot fuse act-order projections with different
    g_idx on its exllama path, :176-183; its cuda path carries a 3K-long g_idx, qlinear_cuda.py:300-312).  The host model's
    attention code is untouched, where the reference replaces the whole module (fused_llama_attn.py:18-135)."""

    def __init__(self, state: _FusedQKVState, index: int):
        super().__init__()
        self.index = index
        self._state = [state]                       # in a list: the shared state is not a submodule
        self.proj = state.layers[index]             # each stand-in owns its own projection: .to() / state_dict() keep working

    @staticmethod
    def _key(x):
        # Inference tensors (torch.inference_mode(): what the reference's generate() runs under, auto_gptq/modeling/_base.py:415-418) do not
        # track a version counter -- reading ._version raises there.  They cannot be modified in place outside inference mode either, so
        # storage + shape + dtype identifies the activation for the three back-to-back projection calls.
        ver = None if x.is_inference() else x._version
        return (x.data_ptr(), tuple(x.shape), x.dtype, ver)

    def forward(self, x):
        from .qlinear_mi355x import forward_multi
        st = self._state[0]
        if self.index == 0:
            st.src = self._key(x)
            st.parts = forward_multi(st.layers, x)
            return st.parts[0]
        # k_proj / v_proj: the same activation as q_proj saw?  Compared by storage, shape, dtype and version counter, not by object identity:
        # wrappers that re-wrap or move the input per call (accelerate's AlignDevicesHook, autocast) hand every projection its own tensor object.
        same = st.parts is not None and st.src == self._key(x)
        if not same:                                # a different input (or q_proj was never called): this projection on its own
            st.src = st.parts = None
            return self.proj(x)
        out = st.parts[self.index]
        if self.index == 2:
            st.src = st.parts = None                # do not keep activations alive between calls
        return out


class FusedGateUpMLP(nn.Module):
    """down(silu(gate(x)) * up(x)) (the role of FusedLlamaMLPForQuantizedModel, auto_gptq/nn_modules/fused_llama_mlp.py:131-306).
    Three mi355x QuantLinears: ONE C-ABI call (gptq_mlp_forward: gate and up in one launch for decode rows, SiLU*mul on fp32, down) over the
    three checkpoint layers as they are -- no concatenated copy of the packed tensors (the reference builds one), so nothing is held twice and
    per-projection act-order is allowed.  A non-quantized down projection: the [gate | up] layer with the SiLU*mul epilogue when gate and up
    share g_idx, else gate and up through ``forward_multi`` and an elementwise SiLU*mul."""

    def __init__(self, gate: QuantLinear, up: QuantLinear, down: nn.Module):
        super().__init__()
        self.gate_up = None
        self.gate_proj, self.up_proj = gate, up
        if not isinstance(down, QuantLinear):
            try:
                self.gate_up = fuse_gate_up(gate, up).to(gate.qweight.device)
                self.gate_proj = self.up_proj = None
            except ValueError:                      # per-projection act-order
                pass
        self.down_proj = down

    def forward(self, x):
        if self.gate_up is not None:
            return self.down_proj(self.gate_up(x))
        if isinstance(self.down_proj, QuantLinear):               # three mi355x layers: one C-ABI call (gptq_mlp_forward)
            from .qlinear_mi355x import mlp_forward
            return mlp_forward(self.gate_proj, self.up_proj, self.down_proj, x)
        from .qlinear_mi355x import forward_multi
        g, u = forward_multi([self.gate_proj, self.up_proj], x)
        return self.down_proj(torch.nn.functional.silu(g) * u)


def inject_fused_llama(model: nn.Module, fuse_attention: bool = True, fuse_mlp: bool = True) -> int:
    """Fuse q/k/v and gate/up of every Llama-style decoder block whose projections are mi355x QuantLinears (the reference's
    inject_to_model entry points, fused_llama_attn.py:137-231 / fused_llama_mlp.py:290-306).  Returns the number of fused modules."""
    n = 0
    for mod in list(model.modules()):
        if fuse_attention and all(isinstance(getattr(mod, a, None), QuantLinear) for a in ("q_proj", "k_proj", "v_proj")):
            q, k, v = mod.q_proj, mod.k_proj, mod.v_proj
            if q.infeatures == k.infeatures == v.infeatures and q.scales.dtype == k.scales.dtype == v.scales.dtype:
                st = _FusedQKVState((q, k, v))
                mod.q_proj, mod.k_proj, mod.v_proj = _QKVPart(st, 0), _QKVPart(st, 1), _QKVPart(st, 2)
                n += 1
        if fuse_mlp and all(isinstance(getattr(mod, a, None), QuantLinear) for a in ("gate_proj", "up_proj")) and hasattr(mod, "down_proj") \
                and getattr(getattr(mod, "act_fn", None), "__class__", type(None)).__name__ in ("SiLU", "SiLUActivation"):
            gate, up = mod.gate_proj, mod.up_proj
            if gate.infeatures == up.infeatures and gate.outfeatures == up.outfeatures and gate.bits == up.bits:
                fm = FusedGateUpMLP(gate, up, mod.down_proj)
                mod.forward = fm.forward
                if fm.gate_up is None:
                    # the fused caller only REFERENCES the three projections: it is kept out of the module tree (object.__setattr__), they stay
                    # registered once under their checkpoint names, and state_dict() of a fused model round-trips with the checkpoint layout
                    object.__setattr__(mod, "fused_mlp", fm)
                else:                               # non-quantized down: the concatenated [gate | up] copy replaces the two originals
                    mod.fused_mlp = fm
                    del mod.gate_proj, mod.up_proj
                n += 1
    return n


__all__ = ["fuse_quant_linears", "fuse_qkv", "fuse_gate_up", "inject_fused_llama", "FusedGateUpMLP"]
