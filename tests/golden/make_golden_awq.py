#!/usr/bin/env python3
"""Golden fixtures for the AWQ -> GPTQ ingest path, produced by the REFERENCE's own functions.

``auto_gptq/modeling/_utils.py`` cannot be imported here (package-relative imports; ``import auto_gptq`` fails under
transformers 5.x), so the three functions on the path -- ``awq_reverse_reorder_int_tensor`` (:525-553),
``unpack_awq`` (:556-621), ``pack_from_tensors`` (:624-701) -- are located with ``ast`` and executed from the reference
file where it lies; nothing of their text is stored in this repository.  They call ``.cuda()`` on their inputs; there is no
GPU in the build container, so ``torch.Tensor.cuda`` is replaced by the identity for the duration of the run (the
arithmetic is plain ATen int/half ops, identical on CPU: every half op is computed in fp32 and rounded once).

AWQ checkpoint side (input): ``qweight int32 [K, N/8]`` (nibble p of word c = column 8c + [0,2,4,6,1,3,5,7][p]), ``qzeros
int32 [G, N/8]`` (same order, raw zero-point), ``scales fp16 [G, N]``.  Outputs stored: the intermediate ``fp16_weight [N, K]``
and ``zeros int8 [G, N]`` of ``unpack_awq``, and the GPTQ ``qweight int32 [K/8, N]`` / ``qzeros int32 [G, N/8]`` of
``pack_from_tensors``.

Usage:  python tests/golden/make_golden_awq.py
"""
import ast
import os
import sys

import numpy as np
import torch

REF = os.environ.get("GPTQ_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
AWQ_ORDER = [0, 2, 4, 6, 1, 3, 5, 7]          # AutoAWQ's pack order (the constant the reference undoes, _utils.py:533)


def load_reference_functions():
    path = os.path.join(REF, "auto_gptq/modeling/_utils.py")
    src = open(path).read()
    tree = ast.parse(src)
    wanted = ("awq_reverse_reorder_int_tensor", "unpack_awq", "pack_from_tensors")
    ns = {"torch": torch, "np": np}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in wanted:
            code = compile(ast.Module(body=[node], type_ignores=[]), path, "exec")
            exec(code, ns)
    assert all(w in ns for w in wanted)
    return ns


def awq_pack(vals):
    """[R, N] integers in 0..15 -> AWQ words [R, N/8] (independent of the reference: AutoAWQ's documented order)."""
    R, N = vals.shape
    v = vals.reshape(R, N // 8, 8).astype(np.uint32)
    out = np.zeros((R, N // 8), dtype=np.uint32)
    for p in range(8):
        out |= v[:, :, AWQ_ORDER[p]] << np.uint32(4 * p)
    return out.view(np.int32)


def case(ns, name, K, N, gs, seed, zero_edge=False):
    rng = np.random.default_rng(seed)
    G = K // gs
    w = rng.integers(0, 16, size=(K, N))
    z = rng.integers(0, 16, size=(G, N))
    if zero_edge:
        z[0, :8] = 0                             # zero-point 0 -> GPTQ field (0 - 1) & 15 = 15
        z[-1, -8:] = 15
    scales = torch.from_numpy((0.002 * (1 + rng.random((G, N)))).astype(np.float16))
    awq_qweight = torch.from_numpy(awq_pack(w))
    awq_qzeros = torch.from_numpy(awq_pack(z))
    fp16_weight, zeros = ns["unpack_awq"](awq_qweight, awq_qzeros, scales, 4, gs)
    qweight, qzeros = ns["pack_from_tensors"](fp16_weight, zeros, scales, 4, gs)
    np.savez_compressed(os.path.join(HERE, f"awq_{name}.npz"), awq_qweight=awq_qweight.numpy(), awq_qzeros=awq_qzeros.numpy(),
                        scales=scales.numpy(), w=w.astype(np.uint8), z=z.astype(np.uint8),
                        fp16_weight=fp16_weight.contiguous().numpy(), zeros=zeros.numpy(),
                        qweight=qweight.numpy(), qzeros=qzeros.numpy(), K=K, N=N, group_size=gs)
    print(f"awq_{name}: fp16_weight {tuple(fp16_weight.shape)} {fp16_weight.dtype} zeros {tuple(zeros.shape)} {zeros.dtype} "
          f"qweight {tuple(qweight.shape)} qzeros {tuple(qzeros.shape)}")


def main():
    ns = load_reference_functions()
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        case(ns, "k128_n64_g32", 128, 64, 32, 0)
        case(ns, "k256_n128_g128", 256, 128, 128, 1)   # the reference reshapes through (-1, group_size, K): N % group_size must be 0
        case(ns, "k64_n64_g32_zero_edges", 64, 64, 32, 2, zero_edge=True)
    finally:
        torch.Tensor.cuda = orig


if __name__ == "__main__":
    sys.exit(main())
