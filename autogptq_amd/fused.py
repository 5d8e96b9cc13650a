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


__all__ = ["fuse_quant_linears", "fuse_qkv", "fuse_gate_up"]
