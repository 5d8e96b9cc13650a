"""End-to-end through the callers either side of the hot path: tiny random Llama -> quantise -> pack_model ->
GPTQ-v1 checkpoint on disk -> fresh skeleton -> load_packed_layers -> autogptq_post_init -> (fused injectors) -> generate.

The reference's own end-to-end test has this shape (tests/test_q4.py:1165-1222: from_quantized + generate, compared with a
known-good text); with no network the "known good" here is the fp16 twin whose linears hold the dequantised weights.
CPU part: the checkpoint round trip and the attribute contract against the reference's GeneralQuantLinear shim.
GPU part (-m gpu): logits and greedy tokens of the quantized model (HIP kernels) against the twin.
"""
import importlib.util
import os

import pytest
import torch

import _tiny_llama as TL

REF_SHIM = "/root/reference/auto_gptq/nn_modules/qlinear/__init__.py"


@pytest.mark.parametrize("desc_act", [False, True], ids=["seq", "act"])
def test_checkpoint_round_trip_cpu(tmp_path, desc_act):
    """save -> load_packed_layers reproduces every tensor of the quantized model (packed ints bit for bit), on CPU."""
    from autogptq_amd import QuantLinear

    m = TL.fresh_model(1)
    TL.quantize_and_pack(m, desc_act)
    TL.save_checkpoint(m, str(tmp_path), desc_act)
    loaded, sd, qc = TL.load_checkpoint(str(tmp_path))
    assert qc["desc_act"] == desc_act and qc["bits"] == 4
    a, b = m.state_dict(), loaded.state_dict()
    assert set(a) == set(b)
    for k in a:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k].cpu(), b[k].cpu()), k
    n_q = sum(isinstance(x, QuantLinear) for x in loaded.modules())
    assert n_q == 2 * 7                                  # q,k,v,o,gate,up,down per block; lm_head stays fp16
    assert {"qweight", "qzeros", "scales", "g_idx"} <= {k.rsplit(".", 1)[1] for k in sd if "q_proj" in k}


@pytest.mark.skipif(not os.path.exists(REF_SHIM), reason="reference tree not present (GPU box)")
def test_reference_general_quant_linear_shim_accepts_this_backend():
    """auto_gptq/nn_modules/qlinear/__init__.py:4-56 wraps ANY backend's QuantLinear by reading its attributes and buffers;
    loaded by path (it only imports torch.nn) and wrapped around this class, on CPU."""
    from autogptq_amd import QuantLinear

    spec = importlib.util.spec_from_file_location("ref_qlinear_shim", REF_SHIM)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    q = QuantLinear(4, 128, 256, 512, True)
    g = mod.GeneralQuantLinear(q)
    assert (g.infeatures, g.outfeatures, g.bits, g.group_size, g.maxq) == (256, 512, 4, 128, 15)
    assert g.in_features == 256 and g.out_features == 512 and g.trainable is False
    for name in ("qweight", "qzeros", "scales", "g_idx"):
        assert getattr(g, name) is getattr(q, name), name            # the shim registers the SAME tensors
    assert g.weight.data_ptr() == q.qweight.data_ptr() and g.bias.data_ptr() == q.bias.data_ptr()
    assert g.kernel_switch_threshold == q.kernel_switch_threshold
    assert g.forward == q.forward                                     # bound method of this backend: forward goes to the HIP path

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = QuantLinear(4, 128, 256, 512, True)
            self.b = torch.nn.Linear(4, 4)
    h = Holder()
    mod.GeneralQuantLinear.inject_to_model(h, QuantLinear)
    assert type(h.a).__name__ == "GeneralQuantLinear" and isinstance(h.b, torch.nn.Linear) and not hasattr(h.b, "qweight")


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True], ids=["plain", "fused"])
@pytest.mark.parametrize("desc_act", [False, True], ids=["seq", "act"])
def test_tiny_llama_generate_matches_dequantised_twin(tmp_path, desc_act, fused):
    from autogptq_amd import QuantLinear
    from autogptq_amd.fused import inject_fused_llama
    from autogptq_amd.model_utils import autogptq_post_init

    dev = "cuda:0"
    m = TL.fresh_model(1)
    twin_w = TL.quantize_and_pack(m, desc_act)
    TL.save_checkpoint(m, str(tmp_path), desc_act)
    twin = TL.make_twin({k: v.cpu() for k, v in m.state_dict().items()}, twin_w).to(dev)
    del m
    qm, _, _ = TL.load_checkpoint(str(tmp_path))
    qm = qm.to(dev)
    if fused:
        n = inject_fused_llama(qm)
        assert n == 2 * 2                               # attention + MLP of both blocks, act-order included (no tensor concatenation needed)
    qm = autogptq_post_init(qm, use_act_order=desc_act, max_input_length=64)
    assert any(isinstance(x, QuantLinear) for x in qm.modules())

    ids = torch.randint(0, 512, (1, 12), generator=torch.Generator().manual_seed(5)).to(dev)
    with torch.no_grad():
        lt = twin(ids).logits.float()
        lq = qm(ids).logits.float()
    scale = float(lt.abs().max())
    assert float((lq - lt).abs().max()) <= 2e-2 * scale, (float((lq - lt).abs().max()), scale)
    with torch.no_grad():
        gt = twin.generate(ids, max_new_tokens=16, do_sample=False, pad_token_id=0)
        gq = qm.generate(ids, max_new_tokens=16, do_sample=False, pad_token_id=0)
    assert gt.shape == (1, 28)
    assert torch.equal(gt, gq), (gt.tolist(), gq.tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("desc_act", [False, True], ids=["seq", "act"])
def test_tiny_llama_decode_step_as_one_hipgraph(tmp_path, desc_act):
    """model_utils.capture_decode_step: one decode step of the quantised tiny Llama with a StaticCache as ONE hipGraph -- replayed
    token by token it must produce exactly the tokens the same model generates eagerly (greedy), and its logits must match the eager step's."""
    from transformers import StaticCache
    from autogptq_amd.fused import inject_fused_llama
    from autogptq_amd.model_utils import autogptq_post_init, capture_decode_step

    dev = "cuda:0"
    m = TL.fresh_model(2)
    TL.quantize_and_pack(m, desc_act)
    TL.save_checkpoint(m, str(tmp_path), desc_act)
    del m
    qm, _, _ = TL.load_checkpoint(str(tmp_path))
    qm = qm.to(dev)
    inject_fused_llama(qm)                                  # q|k|v and gate|up through gptq_forward_multi: the grouped launches of bench.py's headline
    qm = autogptq_post_init(qm, use_act_order=desc_act, max_input_length=64)
    qm.set_attn_implementation("sdpa")                      # transformers' eager mask path builds torch.tensor(0.0, device=...) per call: a host copy, not capturable
    P, NEW, L = 9, 12, 32
    ids = torch.randint(0, 512, (1, P), generator=torch.Generator().manual_seed(7)).to(dev)
    with torch.no_grad():
        want = qm.generate(ids, max_new_tokens=NEW, do_sample=False, pad_token_id=0)[0, P:].tolist()
        cache = StaticCache(qm.config, max_cache_len=L)
        logits = qm(ids, past_key_values=cache, use_cache=True).logits
        step = capture_decode_step(qm, cache)
        tok = logits[:, -1].argmax(-1)
        got = [int(tok)]
        for i in range(NEW - 1):
            lg = step(tok.view(1, 1))
            if i == 0:                                      # the captured step against the same step run eagerly on a second cache
                c2 = StaticCache(qm.config, max_cache_len=L)
                qm(ids, past_key_values=c2, use_cache=True)
                le = qm(tok.view(1, 1), past_key_values=c2, use_cache=True).logits
                assert float((lg.float() - le.float()).abs().max()) <= 2e-3 * float(le.float().abs().max())
            tok = lg[:, -1].argmax(-1)
            got.append(int(tok))
    assert got == want, (got, want)
