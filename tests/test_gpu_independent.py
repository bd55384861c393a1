"""The HIP path against an implementation written by other people, with no arithmetic of this repository's oracle in
between: the base LCM's frontend -> pre-norm RMSNorm / RoPE / SwiGLU causal decoder -> postnet
(lcm/models/base_lcm/builder.py:94-124) is, layer for layer, the architecture of Hugging Face `transformers`'
LlamaModel (installed in the image), so the same weights go into both.  The rotary pairing differs by convention
(consecutive pairs in fairseq2 and the original LLaMA code, half against half in transformers) and is bridged by the
row permutation of W_q / W_k that transformers' own LLaMA checkpoint converter applies."""

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _config(name):
    from lcm.models.base_lcm import BaseLCModelConfig, TransformerConfig
    if name == "mini":
        return BaseLCModelConfig(
            model_dim=256, max_seq_len=512, sonar_normalizer_name="dummy_sonar_normalizer",
            lcm=TransformerConfig(num_layers=3, num_attn_heads=2, ffn_inner_dim=1024, use_swiglu=True,
                                  layer_normalization_style="rms", pos_embedding_style="rope", mha_output_proj_bias=True))
    if name == "big":                  # the 1.6B arch's widths on two layers: with 8 x 128 positions the matrix-core products
        return BaseLCModelConfig(      # and the tiled prefill attention are the kernels that run
            model_dim=2048, max_seq_len=512, sonar_normalizer_name="dummy_sonar_normalizer",
            lcm=TransformerConfig(num_layers=2, num_attn_heads=16, ffn_inner_dim=8192, use_swiglu=True,
                                  layer_normalization_style="rms", pos_embedding_style="rope", mha_output_proj_bias=True))
    return BaseLCModelConfig(          # "wide": 8 heads, the GLU width rounding of a 4096-wide FFN
        model_dim=1024, max_seq_len=512, sonar_normalizer_name="dummy_sonar_normalizer",
        lcm=TransformerConfig(num_layers=2, num_attn_heads=8, ffn_inner_dim=4096, use_swiglu=True,
                              layer_normalization_style="rms", pos_embedding_style="rope", mha_output_proj_bias=True))


def _llama_for(cfg, sd, tower="lcm"):
    transformers = pytest.importorskip("transformers")
    tc = cfg.lcm if tower == "lcm" else cfg.context_encoder
    H, d = tc.num_attn_heads, cfg.model_dim
    hf_cfg = transformers.LlamaConfig(
        vocab_size=8, hidden_size=d, intermediate_size=sd[f"{tower}.layers.0.ffn.inner_proj.weight"].shape[0],
        num_hidden_layers=tc.num_layers, num_attention_heads=H, num_key_value_heads=H, rms_norm_eps=1e-5,
        rope_theta=10000.0, attention_bias=True, mlp_bias=True, max_position_embeddings=256, attn_implementation="eager")
    hf = transformers.LlamaModel(hf_cfg).eval()

    def halves_first(w):        # rows (head, pair, 2) -> (head, 2, pair)
        return w.reshape(H, d // H // 2, 2, *w.shape[1:]).transpose(1, 2).reshape(w.shape)

    with torch.no_grad():
        for i, layer in enumerate(hf.layers):
            p, a, m = f"{tower}.layers.{i}", layer.self_attn, layer.mlp
            for dst, src, perm in ((a.q_proj, "self_attn.q_proj", True), (a.k_proj, "self_attn.k_proj", True),
                                   (a.v_proj, "self_attn.v_proj", False), (a.o_proj, "self_attn.output_proj", False),
                                   (m.gate_proj, "ffn.gate_proj", False), (m.up_proj, "ffn.inner_proj", False),
                                   (m.down_proj, "ffn.output_proj", False)):
                w, b = sd[f"{p}.{src}.weight"], sd[f"{p}.{src}.bias"]
                dst.weight.copy_(halves_first(w) if perm else w)
                dst.bias.copy_(halves_first(b) if perm else b)
            layer.input_layernorm.weight.copy_(sd[p + ".self_attn_layer_norm.weight"])
            layer.post_attention_layernorm.weight.copy_(sd[p + ".ffn_layer_norm.weight"])
        hf.norm.weight.copy_(sd[f"{tower}.layer_norm.weight"])
    return hf


def _draw(model, g):
    """Weights drawn here, by name and shape of the model's own state dict (norm weights away from 1, biases away from 0)."""
    fresh = {}
    for k, t in model.state_dict().items():
        if "sonar_normalizer" in k:
            fresh[k] = t.detach().float().cpu()
        elif k.endswith("layer_norm.weight"):
            fresh[k] = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
        elif k.endswith(".bias"):
            fresh[k] = 0.05 * torch.randn(t.shape, generator=g)
        else:
            fresh[k] = torch.randn(t.shape, generator=g) * (2.0 / t.shape[-1] ** 0.5 if "q_proj" in k or "k_proj" in k
                                                            else 1.0 / t.shape[-1] ** 0.5)
    return fresh


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_tower_contextualizer_against_transformers_llama(dtype):
    """`TwoTowerDiffusionLCModel.encode` (reference builder.py:161-191: frontend + contextualizer on normalized inputs)
    against LlamaModel on the same weights: at once without a state bag, and position by position with one."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.two_tower_diffusion_lcm import (DenoiserConfig, TransformerConfig, TwoTowerDiffusionLCModelConfig,
                                                    create_two_tower_diffusion_lcm_model)
    from lcm.nn.incremental_state import LCMIncrementalStateBag
    from lcm.nn.schedulers import DDIMSchedulerConfig
    shared = dict(num_attn_heads=4, ffn_inner_dim=2048, use_swiglu=True, layer_normalization_style="rms",
                  mha_output_proj_bias=True)
    cfg = TwoTowerDiffusionLCModelConfig(
        model_dim=512, max_seq_len=512,
        context_encoder=TransformerConfig(num_layers=3, pos_embedding_style="rope", **shared),
        denoiser=DenoiserConfig(num_layers=1, pos_embedding_style="none", timestep_embed_dim=512, **shared),
        noise_scheduler=DDIMSchedulerConfig(num_diffusion_train_steps=100),
        sonar_normalizer_name="dummy_sonar_normalizer", trained_with_cf_guidance=True)
    model = create_two_tower_diffusion_lcm_model(cfg, device="cuda:0", dtype=dtype, seed=0)
    g = torch.Generator().manual_seed(23)
    model.load_state_dict(_draw(model, g))
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    hf = _llama_for(cfg, sd, tower="context_encoder")
    x = torch.randn(2, 9, 1024, generator=g)
    with torch.no_grad():
        want = hf(inputs_embeds=F.linear(x.to(dtype).float(), sd["encoder_frontend.pre_linear.weight"],
                                         sd["encoder_frontend.pre_linear.bias"])).last_hidden_state

    def check(got):
        got = got.float().cpu()
        assert got.shape == want.shape and torch.isfinite(got).all()
        cos = float(F.cosine_similarity(got.flatten(0, 1), want.flatten(0, 1), dim=-1).min())
        if dtype == torch.float32:
            assert float((got - want).abs().max()) <= 1e-3 * max(1.0, float(want.abs().max())) and cos >= 0.9999, cos
        else:
            assert cos >= 0.999, cos

    out = model.encode(EmbeddingsBatch(x.cuda().to(dtype), None))
    check(out.seqs if hasattr(out, "seqs") else out)


@pytest.mark.parametrize("name,dtype", [("mini", torch.float32), ("wide", torch.float32), ("mini", torch.bfloat16),
                                        ("big", torch.bfloat16), ("big", torch.float32)])
def test_base_lcm_forward_against_transformers_llama(name, dtype):
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.base_lcm import create_base_lcm_model
    from lcm.nn.incremental_state import LCMIncrementalStateBag

    cfg = _config(name)
    tc = cfg.lcm
    assert tc.use_swiglu and tc.layer_normalization_style == "rms" and tc.pos_embedding_style == "rope", tc
    model = create_base_lcm_model(cfg, device="cuda:0", dtype=dtype, seed=0)
    # what the device holds (rounded to its dtype) is what the other implementation gets, in fp32
    g = torch.Generator().manual_seed(17)
    fresh = _draw(model, g)
    model.load_state_dict(fresh)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    hf = _llama_for(cfg, sd)

    shape = (8, 128) if name == "big" else (3, 11)
    x = torch.randn(*shape, cfg.sonar_embed_dim, generator=g)         # already-normalized inputs
    got = model.forward(EmbeddingsBatch(x.cuda().to(dtype), None)).seqs.float().cpu()
    xin = x.to(dtype).float()
    scale = cfg.model_dim ** 0.5 if cfg.frontend.scale_embeddings else 1.0
    with torch.no_grad():
        hidden = hf(inputs_embeds=F.linear(scale * xin, sd["frontend.pre_linear.weight"], sd["frontend.pre_linear.bias"]))
        want = F.linear(hidden.last_hidden_state, sd["postnet.fc.weight"], sd["postnet.fc.bias"])
    assert got.shape == want.shape and torch.isfinite(got).all()
    err = float((got - want).abs().max())
    cos = float(F.cosine_similarity(got.flatten(0, 1), want.flatten(0, 1), dim=-1).min())
    if dtype == torch.float32:
        assert err <= 1e-3 * max(1.0, float(want.abs().max())) and cos >= 0.9999, (err, cos)
    else:
        assert cos >= 0.999, (err, cos)

    if name == "big":
        return
    # position by position through the KV cache (the generators' path): the same answers
    bag = LCMIncrementalStateBag(16)
    steps = []
    for j in range(x.shape[1]):
        steps.append(model.forward(EmbeddingsBatch(x[:, j:j + 1].cuda().to(dtype), None), state_bag=bag).seqs.float().cpu())
        bag.increment_step_nr(1)
    inc = torch.cat(steps, dim=1)
    cos_inc = float(F.cosine_similarity(inc.flatten(0, 1), want.flatten(0, 1), dim=-1).min())
    if dtype == torch.float32:
        assert float((inc - want).abs().max()) <= 1e-3 * max(1.0, float(want.abs().max())) and cos_inc >= 0.9999
    else:
        assert cos_inc >= 0.999, cos_inc
