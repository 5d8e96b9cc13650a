"""Shared by the CPU and GPU end-to-end tests: a randomly initialised tiny Llama, quantised with the oracle's min/max
quantizer, packed by this backend's pack_model, saved as a GPTQ-v1 checkpoint (safetensors + quantize_config.json) -- the shape
of what AutoGPTQ's save_quantized writes (auto_gptq/modeling/_base.py:522-615) -- and its dequantised fp16 twin."""
import json
import os

import torch

from oracle import gptq_oracle as O

BITS, GROUP = 4, 64


def tiny_config():
    from transformers import LlamaConfig
    return LlamaConfig(hidden_size=256, intermediate_size=704, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
                       vocab_size=512, max_position_embeddings=128, attn_implementation="eager", tie_word_embeddings=False)


def fresh_model(seed):
    from transformers import LlamaForCausalLM
    torch.manual_seed(seed)
    m = LlamaForCausalLM(tiny_config())
    m.lm_head.weight.data.normal_(0, 0.3)          # spread the logits: greedy argmax is then far from ties
    return m.half().eval()


def quantizable(model):
    from autogptq_amd.model_utils import find_layers
    return {n: l for n, l in find_layers(model).items() if n.startswith("model.layers.")}


def quantize_and_pack(model, desc_act, seed=0):
    """Oracle min/max quantizer per (group, column) -> quantizers dict -> pack_model.  Returns the twin's weights
    {name: dequantised [N, K] fp16} computed by the ORACLE from the packed tensors."""
    from autogptq_amd.model_utils import pack_model

    gen = torch.Generator().manual_seed(seed)
    quantizers = {}
    for name, lin in quantizable(model).items():
        K = lin.in_features
        gi = torch.from_numpy(O.default_g_idx(K, GROUP))
        if desc_act:
            gi = gi[torch.randperm(K, generator=gen)].contiguous()
        s, z = O.minmax_quantize(lin.weight.data.float(), BITS, GROUP, g_idx=gi.numpy())
        quantizers[name] = (None, s.half(), z.half(), gi)
    pack_model(model, quantizers, BITS, GROUP, desc_act=desc_act)
    twin_w = {}
    mode = O.reference_zero_mode(desc_act, BITS)
    for name in quantizers:
        q = model.get_submodule(name)
        W = O.dequantize(q.qweight.cpu(), q.qzeros.cpu(), q.scales.cpu(), q.g_idx.cpu(), BITS, mode)     # [K, N]
        twin_w[name] = W.t().contiguous()
    return twin_w


def save_checkpoint(model, path, desc_act):
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    sd = {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    save_file(sd, os.path.join(path, "model.safetensors"), metadata={"format": "pt"})
    with open(os.path.join(path, "quantize_config.json"), "w") as f:     # keys of BaseQuantizeConfig.to_dict (quantization/config.py:263-283)
        json.dump({"bits": BITS, "group_size": GROUP, "damp_percent": 0.01, "desc_act": desc_act, "static_groups": False, "sym": False,
                   "true_sequential": True, "model_name_or_path": None, "model_file_base_name": "model", "quant_method": "gptq",
                   "checkpoint_format": "gptq"}, f)


def load_checkpoint(path, seed=99):
    """The tensor-facing part of from_quantized (auto_gptq/modeling/_base.py:691-1247): skeleton with unrelated random weights ->
    swap the quantized linears -> fill everything from the file."""
    from safetensors.torch import load_file
    from autogptq_amd.model_utils import load_packed_layers

    with open(os.path.join(path, "quantize_config.json")) as f:
        qc = json.load(f)
    sd = load_file(os.path.join(path, "model.safetensors"))
    model = fresh_model(seed)
    return load_packed_layers(model, sd, qc["bits"], qc["group_size"], desc_act=qc["desc_act"], quant_method=qc["quant_method"],
                              checkpoint_format=qc["checkpoint_format"]), sd, qc


def make_twin(src_model_state, twin_w, seed=7):
    """fp16 model whose linears hold the dequantised weights and whose other tensors equal the quantized model's."""
    twin = fresh_model(seed)
    sd = twin.state_dict()
    for k in sd:
        if k in src_model_state and src_model_state[k].shape == sd[k].shape and src_model_state[k].dtype == sd[k].dtype:
            sd[k] = src_model_state[k].clone()
    for name, W in twin_w.items():
        sd[name + ".weight"] = W.clone()
    twin.load_state_dict(sd)
    return twin
