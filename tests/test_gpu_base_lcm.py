"""Parity of the base (MSE) LCM on the HIP path against the CPU oracle: forward over whole sequences
(the base-LCM benchmark shape, reduced) and sentence-by-sentence generation with the KV cache.

Tolerances as for the two-tower path: fp32 max-abs <= 1e-3 (normalizer units) and cosine >= 0.9999; bf16
cosine >= 0.999.  The reference's own invariants for this generator are restated on the HIP path:
KV cache on == off (tests/units/inference/test_base_lcm_kv_caching.py, atol 1e-3) and batch == singles
(tests/units/inference/test_base_lcm_batched_inference.py).
"""

import pytest
import torch

import oracle
from parity_utils import (base_config, compare, compare_raw, make_base_problem, run_hip_base, run_oracle_base)

pytestmark = pytest.mark.gpu

FP32_MAX_ABS, FP32_COS, BF16_COS = 1e-3, 0.9999, 0.999
OPTS = dict(eos_threshold=None)


def _norm_sd(sd):
    return {"sonar_normalizer.center": sd["frontend.sonar_normalizer.center"],
            "sonar_normalizer.scale": sd["frontend.sonar_normalizer.scale"]}


@pytest.mark.parametrize("sample", [True, False])
def test_toy_base_generation_fp32(sample):
    """toy_base_lcm (LayerNorm, learned positions, ReLU FFN, no normalizer): 4 docs x (3 + 5)."""
    arch = "toy_base_lcm"
    cfg, sd, x, nz = make_base_problem(arch, 4, 3, 5)
    opts = dict(OPTS, sample_latent_variable=sample, lcm_temperature=0.3)
    ref = run_oracle_base(cfg, sd, x, nz, 5, opts)
    out, _ = run_hip_base(arch, sd, x, nz, 5, opts)
    worst, cos = compare_raw(out.hypotheses, ref, start=3)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)
    assert all(h[0].seq.shape == (8, 1024) for h in out.hypotheses)


def test_mini_base_generation_fp32_and_bf16():
    """1.6B-style blocks (RMSNorm, RoPE, SwiGLU, out-proj bias, normalizer)."""
    arch = "mini_base_lcm"
    cfg, sd, x, nz = make_base_problem(arch, 5, 4, 6)
    opts = dict(OPTS, lcm_temperature=0.5)
    ref = run_oracle_base(cfg, sd, x, nz, 6, opts)
    out, _ = run_hip_base(arch, sd, x, nz, 6, opts)
    worst, cos = compare(out.hypotheses, ref, _norm_sd(sd), start=4)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)
    sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
    ref16 = run_oracle_base(cfg, sd16, x.bfloat16().float(), nz, 6, opts)
    out16, _ = run_hip_base(arch, sd, x, nz, 6, opts, dtype=torch.bfloat16)
    worst, cos = compare(out16.hypotheses, ref16, _norm_sd(sd16), start=4)
    assert cos >= BF16_COS, (worst, cos)


@pytest.mark.parametrize("arch,B,S", [("toy_base_lcm", 3, 40), ("mini_base_lcm", 6, 200)])
def test_base_forward_all_positions(arch, B, S):
    """predict_next_sentence over whole sequences without a cache (the benchmark forward's shape, reduced): rows
    B * S exceed 256, so the bf16 run goes through the many-row matrix-core GEMM."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.base_lcm import create_base_lcm_model
    cfg, sd, x, _ = make_base_problem(arch, B, S, 0)
    ref = oracle.OracleBaseLCM(cfg, sd).predict(x)
    model = create_base_lcm_model(base_config(arch), device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    out = model.predict_next_sentence(EmbeddingsBatch(x.cuda(), None)).seqs.cpu()
    scale = ref.abs().max()
    assert (out - ref).abs().max() <= 2e-4 * scale
    assert torch.nn.functional.cosine_similarity(out, ref, dim=-1).min() >= FP32_COS
    if arch == "mini_base_lcm":
        sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
        ref16 = oracle.OracleBaseLCM(cfg, sd16).predict(x.bfloat16().float())
        m16 = create_base_lcm_model(base_config(arch), device="cuda:0", dtype=torch.bfloat16)
        m16.load_state_dict(sd)
        out16 = m16.predict_next_sentence(EmbeddingsBatch(x.cuda().bfloat16(), None)).seqs.cpu()
        c, s = sd16["frontend.sonar_normalizer.center"], sd16["frontend.sonar_normalizer.scale"]
        cos = torch.nn.functional.cosine_similarity((out16 - c) / s, (ref16 - c) / s, dim=-1)
        assert cos.min() >= BF16_COS, float(cos.min())


def test_kv_cache_on_equals_off():
    arch = "mini_base_lcm"
    cfg, sd, x, nz = make_base_problem(arch, 3, 4, 4)
    opts = dict(OPTS, lcm_temperature=0.5)
    a, model = run_hip_base(arch, sd, x, nz, 4, opts)
    b, _ = run_hip_base(arch, sd, x, nz, 4, opts, model=model, disable_cache=True)
    for ha, hb in zip(a.hypotheses, b.hypotheses):
        assert torch.allclose(ha[0].seq, hb[0].seq, atol=1e-3, rtol=1e-3)


def test_batch_equals_singles_and_ragged_eos():
    arch = "mini_base_lcm"
    lens = [2, 4, 3]
    cfg, sd, x, nz = make_base_problem(arch, 3, 4, 4, prompt_lens=lens)
    opts = dict(OPTS, lcm_temperature=0.5)
    ref = run_oracle_base(cfg, sd, x, nz, 4, opts, prompt_lens=lens)
    out, model = run_hip_base(arch, sd, x, nz, 4, opts, prompt_lens=lens)
    for i, n in enumerate(lens):
        assert out.hypotheses[i][0].seq.shape[0] == n + 4 == ref[i][0].seq.shape[0]
    worst, cos = compare(out.hypotheses, ref, _norm_sd(sd))
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)
    # EOS: stop document 1 at its second generated sentence
    eos = ref[1][0].seq[5].clone()
    opts_eos = dict(opts, eos_threshold=0.999, trim_hypotheses=True)
    ref_e = run_oracle_base(cfg, sd, x, nz, 4, opts_eos, prompt_lens=lens, eos_vec=eos)
    out_e, _ = run_hip_base(arch, sd, x, nz, 4, opts_eos, prompt_lens=lens, eos_vec=eos, model=model)
    assert [h[0].seq.shape[0] for h in out_e.hypotheses] == [h[0].seq.shape[0] for h in ref_e]
    assert ref_e[1][0].seq.shape[0] == 5
    # a document alone == the same document in the batch (unpadded prompts, noise indexed by document)
    cfg, sd, x, nz = make_base_problem(arch, 3, 3, 3)
    full, model = run_hip_base(arch, sd, x, nz, 3, opts)
    for i in range(3):
        one, _ = run_hip_base(arch, sd, x[i:i + 1], nz[:, i:i + 1], 3, opts, model=model)
        a, b = one.hypotheses[0][0].seq, full.hypotheses[i][0].seq
        assert (a - b).abs().max() <= 1e-6 * max(1.0, float(b.abs().max()))


def test_base_qk_layernorm_forward_and_generation():
    """`enable_qk_layernorm` on the base LCM's decoder (lcm/nn/transformer/attention.py:143-199; RMSNorm style, RoPE
    after the norms): the whole-sequence forward (many-row kernels) and cached generation against the oracle, fp32."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.base_lcm import create_base_lcm_model
    from oracle.weights import synthetic_noise, synthetic_prompts
    arch = "mini_base_lcm"
    cfg = oracle.get_base_arch(arch)
    cfg.lcm.qk_layernorm = True
    sd = oracle.init_state_dict(cfg, seed=3, qk_gain=2.0, norm_jitter=0.2)
    assert "lcm.layers.0.self_attn.k_layer_norm.weight" in sd
    mc = base_config(arch)
    mc.lcm.enable_qk_layernorm = True
    model = create_base_lcm_model(mc, device="cuda:0", dtype=torch.float32)
    assert set(model.state_dict()) == set(sd)
    model.load_state_dict(sd)
    x = synthetic_prompts(oracle.load_dummy_normalizer(), 5, 90, seed=77)
    ref = oracle.OracleBaseLCM(cfg, sd).predict(x)
    out = model.predict_next_sentence(EmbeddingsBatch(x.cuda(), None)).seqs.cpu()
    assert (out - ref).abs().max() <= 2e-4 * ref.abs().max()
    assert torch.nn.functional.cosine_similarity(out, ref, dim=-1).min() >= FP32_COS
    nz = synthetic_noise(8, 5, seed=78)
    opts = dict(OPTS, lcm_temperature=0.5)
    ref_g = run_oracle_base(cfg, sd, x[:, :4], nz, 4, opts)
    out_g, _ = run_hip_base(arch, sd, x[:, :4], nz, 4, opts, model=model)
    worst, cos = compare(out_g.hypotheses, ref_g, _norm_sd(sd), start=4)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)


def test_fused_rotation_and_cache_scatter_ragged_rows():
    """The q / k / v product of many rows rotates q / k and writes k / v into the cache in its epilogue.  7 x 300
    positions (2100 rows: the last 128-row tile is ragged; 300 positions per document cross the 64-key stages of the
    prefill attention) on a 2-layer, d = 1024 decoder in bf16: the same bits as with the separate
    rotation-and-scatter pass (sessions created with SESSION_SEPARATE_ROPE), and the bf16 bar against the oracle."""
    from lcm import _hip
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.base_lcm import BaseLCModelConfig, TransformerConfig, create_base_lcm_model
    from oracle.weights import synthetic_prompts
    cfg = oracle.get_base_arch("mini_base_lcm")
    cfg.model_dim, cfg.lcm.num_heads, cfg.lcm.num_layers, cfg.lcm.ffn_inner_dim = 1024, 8, 2, 2048
    sd = oracle.init_state_dict(cfg, seed=5, qk_gain=2.0, pos_std=0.5)
    mc = BaseLCModelConfig(
        model_dim=1024, max_seq_len=512, sonar_normalizer_name="dummy_sonar_normalizer",
        lcm=TransformerConfig(num_layers=2, num_attn_heads=8, ffn_inner_dim=2048, use_swiglu=True,
                              layer_normalization_style="rms", pos_embedding_style="rope", mha_output_proj_bias=True))
    model = create_base_lcm_model(mc, device="cuda:0", dtype=torch.bfloat16)
    assert set(model.state_dict()) == set(sd)
    model.load_state_dict(sd)
    x = synthetic_prompts(oracle.load_dummy_normalizer(), 7, 300, seed=99)
    xb = x.cuda().bfloat16()

    def fwd_and_kernels():
        out = model.predict_next_sentence(EmbeddingsBatch(xb, None)).seqs
        sess = model._session_pool[0]
        sess.profile_enable(True)
        model.predict_next_sentence(EmbeddingsBatch(xb, None))
        names = {r["name"] for r in sess.profile() if r["launches"]}
        sess.profile_enable(False)
        return out, names

    fused, names = fwd_and_kernels()
    assert "ctx.qkv_rope" in names and "ctx.rope_store" not in names, names    # the fused launch is what ran
    model.session_flags = _hip.SESSION_SEPARATE_ROPE
    try:
        separate, names = fwd_and_kernels()
    finally:
        model.session_flags = 0
    assert "ctx.rope_store" in names and "ctx.qkv_rope" not in names, names
    assert torch.equal(fused, separate)
    sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
    ref16 = oracle.OracleBaseLCM(cfg, sd16).predict(x.bfloat16().float())
    c, s = sd16["frontend.sonar_normalizer.center"], sd16["frontend.sonar_normalizer.scale"]
    cos = torch.nn.functional.cosine_similarity((fused.cpu().float() - c) / s, (ref16 - c) / s, dim=-1)
    assert cos.min() >= BF16_COS, float(cos.min())


@pytest.mark.parametrize("arch,sample", [("mini_base_lcm", True), ("mini_base_lcm", False), ("toy_base_lcm", True)])
def test_base_teacher_forced_scorer(arch, sample):
    """LCMScorer (reference lcm/inference/lcm/scorer.py:19-198) against the oracle's restatement of it: ragged
    documents, min_context_len 1 and 3, with and without sampling noise.  Three HIP evaluations must agree with the
    oracle and with each other: all positions in one pass (the default), the reference's step-by-step loop on the KV
    cache, and the loop with the cache disabled."""
    from lcm.datasets.batch import EmbeddingsBatch, PaddingMask
    from lcm.inference.lcm import LCMGeneratorOptions, LCMScorer
    from lcm.models.base_lcm import create_base_lcm_model
    lens = [7, 4, 6, 7, 5]
    cfg, sd, x, nz = make_base_problem(arch, 5, 7, 0, prompt_lens=lens)
    opts = dict(OPTS, sample_latent_variable=sample, lcm_temperature=0.4)
    model = create_base_lcm_model(base_config(arch), device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    batch = EmbeddingsBatch(x.cuda(), PaddingMask(torch.tensor(lens, device="cuda:0"), 7))
    has_norm = "frontend.sonar_normalizer.center" in sd
    for mcl in (1, 3):
        ref = oracle.OracleScorer(oracle.OracleBaseLCM(cfg, sd), oracle.GenOptions(**opts))(
            x, text_lens=torch.tensor(lens), min_context_len=mcl, noise=nz)
        runs = {}
        for name, one_pass, no_cache in (("one pass", True, False), ("steps, KV cache", False, False), ("steps, no cache", False, True)):
            scorer = LCMScorer(model, LCMGeneratorOptions(**opts))
            scorer.one_pass = one_pass
            out = scorer(batch, min_context_len=mcl, disable_cache=no_cache, noise=nz.cuda())
            assert [h[0].seq.shape[0] for h in out.hypotheses] == [n - mcl for n in lens], name
            worst, cos = compare(out.hypotheses, ref, _norm_sd(sd)) if has_norm else compare_raw(out.hypotheses, ref)
            assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (name, mcl, worst, cos)
            runs[name] = out.hypotheses
        for name in ("steps, KV cache", "steps, no cache"):
            for ha, hb in zip(runs["one pass"], runs[name]):
                a, b = ha[0].seq.float(), hb[0].seq.float()
                assert (a - b).abs().max() <= 1e-3 * max(1.0, float(b.abs().max())), name


@pytest.mark.parametrize("order,final,pos", [("post", None, "sine"), ("normformer", "post", "rope")])
def test_base_layer_orders_forward_and_generation(order, final, pos):
    """Post-norm / normformer layers, a stack without its final norm, the learned residual scale and sinusoidal
    positions on the base LCM's decoder (lcm/nn/transformer/factory.py:79-121): whole-sequence forward over 450 rows (the
    many-row kernels) and cached generation against the oracle, fp32; bf16 forward to the bf16 bar."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.base_lcm import create_base_lcm_model
    from oracle.weights import synthetic_noise, synthetic_prompts
    arch = "mini_base_lcm"
    cfg = oracle.get_base_arch(arch)
    cfg.lcm.norm_order, cfg.lcm.final_norm_order, cfg.lcm.pos, cfg.lcm.scale_residual = order, final, pos, True
    sd = oracle.init_state_dict(cfg, seed=5, qk_gain=2.0, norm_jitter=0.2)
    assert "lcm.layer_norm.weight" not in sd and "lcm.layers.2.residual_scale" in sd
    mc = base_config(arch)
    mc.lcm.norm_order_style, mc.lcm.final_norm_order_style, mc.lcm.pos_embedding_style, mc.lcm.scale_residual = order, final, pos, 0.5
    model = create_base_lcm_model(mc, device="cuda:0", dtype=torch.float32)
    assert set(model.state_dict()) == set(sd)
    model.load_state_dict(sd)
    x = synthetic_prompts(oracle.load_dummy_normalizer(), 5, 90, seed=87)
    ref = oracle.OracleBaseLCM(cfg, sd).predict(x)
    out = model.predict_next_sentence(EmbeddingsBatch(x.cuda(), None)).seqs.cpu()
    assert (out - ref).abs().max() <= 2e-4 * ref.abs().max()
    assert torch.nn.functional.cosine_similarity(out, ref, dim=-1).min() >= FP32_COS
    nz = synthetic_noise(8, 5, seed=88)
    opts = dict(OPTS, lcm_temperature=0.5)
    ref_g = run_oracle_base(cfg, sd, x[:, :4], nz, 4, opts)
    out_g, _ = run_hip_base(arch, sd, x[:, :4], nz, 4, opts, model=model)
    worst, cos = compare(out_g.hypotheses, ref_g, _norm_sd(sd), start=4)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)
    sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
    ref16 = oracle.OracleBaseLCM(cfg, sd16).predict(x.bfloat16().float())
    m16 = create_base_lcm_model(mc, device="cuda:0", dtype=torch.bfloat16)
    m16.load_state_dict(sd)
    out16 = m16.predict_next_sentence(EmbeddingsBatch(x.cuda().bfloat16(), None)).seqs.float().cpu()
    assert torch.nn.functional.cosine_similarity(out16, ref16, dim=-1).min() >= BF16_COS


@pytest.mark.parametrize("act", ["tanh", "gelu"])
def test_base_postnet_activation(act):
    """`postnet.activation_name` (lcm/nn/projection.py:73-86): the activation follows the output projection, before the
    sampling noise and the denormalization; forward over all positions and cached generation, fp32."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.base_lcm import create_base_lcm_model
    from oracle.weights import synthetic_noise, synthetic_prompts
    arch = "mini_base_lcm"
    cfg = oracle.get_base_arch(arch)
    cfg.postnet_activation = act
    sd = oracle.init_state_dict(cfg, seed=6, qk_gain=2.0, w_std=0.05)
    mc = base_config(arch)
    mc.postnet.activation_name = act
    model = create_base_lcm_model(mc, device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    x = synthetic_prompts(oracle.load_dummy_normalizer(), 4, 30, seed=97)
    ref = oracle.OracleBaseLCM(cfg, sd).predict(x)
    out = model.predict_next_sentence(EmbeddingsBatch(x.cuda(), None)).seqs.cpu()
    assert (out - ref).abs().max() <= 2e-4 * ref.abs().max()
    cfg.postnet_activation = None
    assert (out - oracle.OracleBaseLCM(cfg, sd).predict(x)).abs().max() > 1e-2 * ref.abs().max()   # it matters here
    cfg.postnet_activation = act
    nz = synthetic_noise(8, 4, seed=98)
    opts = dict(OPTS, lcm_temperature=0.5)
    ref_g = run_oracle_base(cfg, sd, x[:, :4], nz, 4, opts)
    out_g, _ = run_hip_base(arch, sd, x[:, :4], nz, 4, opts, model=model)
    worst, cos = compare(out_g.hypotheses, ref_g, _norm_sd(sd), start=4)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)
    mc.postnet.activation_name = "prelu"
    with pytest.raises(NotImplementedError):
        create_base_lcm_model(mc, device="cuda:0", dtype=torch.float32)


def test_base_prelu_many_rows_bf16():
    """The PReLU epilogue of the many-row matrix-core kernels: toy base LCM (two-matrix FFN) forward over 4 x 160 rows
    in bf16 and fp32 against the oracle."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.base_lcm import create_base_lcm_model, lcm_archs
    from oracle.weights import synthetic_prompts
    arch = "toy_base_lcm"
    cfg = oracle.get_base_arch(arch)
    cfg.lcm.activation = "prelu"
    sd = oracle.init_state_dict(cfg, seed=9, qk_gain=2.0, pos_std=0.5)
    mc = lcm_archs.get_config(arch)
    mc.lcm.ffn_inner_activation_name = "prelu"
    x = synthetic_prompts(oracle.load_dummy_normalizer(), 4, 160, seed=107) * 50.0
    for dtype in (torch.float32, torch.bfloat16):
        sdd = sd if dtype == torch.float32 else {k: v.bfloat16().float() for k, v in sd.items()}
        xd = x if dtype == torch.float32 else x.bfloat16().float()
        ref = oracle.OracleBaseLCM(cfg, sdd).predict(xd)
        model = create_base_lcm_model(mc, device="cuda:0", dtype=dtype)
        assert set(model.state_dict()) == set(sd)
        model.load_state_dict(sd)
        out = model.predict_next_sentence(EmbeddingsBatch(xd.cuda().to(dtype), None)).seqs.float().cpu()
        cos = torch.nn.functional.cosine_similarity(out, ref, dim=-1).min()
        assert cos >= (FP32_COS if dtype == torch.float32 else BF16_COS), (dtype, cos)


def test_option_statements_against_golden_fixture():
    """The committed vectors of tests/golden/option_statements_golden.pt (written by oracle/make_golden_options.py)
    through the HIP path, fp32: layer orders, the stack without a final norm, residual scale, sinusoidal positions, PReLU,
    postnet activation and projections without biases on the base LCM; the packed real FFT on its own.  No oracle
    arithmetic runs here beyond the seeded weight initialisation."""
    import os

    from lcm import _hip
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.base_lcm import create_base_lcm_model
    from oracle import make_golden_options as G
    blob = torch.load(os.path.join(os.path.dirname(__file__), "golden", "option_statements_golden.pt"), weights_only=True)
    lcm_side = {
        "post_sine": lambda mc: (setattr(mc.lcm, "norm_order_style", "post"), setattr(mc.lcm, "pos_embedding_style", "sine")),
        "normformer_nofinal_rope_scaled": lambda mc: (setattr(mc.lcm, "norm_order_style", "normformer"),
                                                      setattr(mc.lcm, "final_norm_order_style", "post"),
                                                      setattr(mc.lcm, "scale_residual", 0.5)),
        "relu_ffn_prelu": lambda mc: (setattr(mc.lcm, "use_swiglu", False), setattr(mc.lcm, "ffn_inner_activation_name", "prelu")),
        "postnet_tanh_no_biases": lambda mc: (setattr(mc.postnet, "activation_name", "tanh"), setattr(mc.postnet, "linear_bias", False),
                                              setattr(mc.frontend, "pre_linear_bias", False)),
    }
    for name, edit in lcm_side.items():
        cfg, sd, x = G.base_case(name)
        mc = base_config("mini_base_lcm")
        edit(mc)
        model = create_base_lcm_model(mc, device="cuda:0", dtype=torch.float32)
        assert set(model.state_dict()) == set(sd), name
        model.load_state_dict(sd)
        case = blob["base"][name]
        out = model.predict_next_sentence(EmbeddingsBatch(case["prompts"].cuda(), None)).seqs.cpu()
        assert (out - case["predict"]).abs().max() <= 2e-4 * case["predict"].abs().max(), name
        assert torch.nn.functional.cosine_similarity(out, case["predict"], dim=-1).min() >= FP32_COS, name
    f = blob["fft"]
    assert (_hip.rfft_pack(f["x"].cuda(), inverse=False).cpu() - f["packed"]).abs().max() <= 5e-6 * f["packed"].abs().max()
    assert (_hip.rfft_pack(f["packed"].cuda(), inverse=True).cpu() - f["back"]).abs().max() <= 5e-6 * f["back"].abs().max()


def _reference_style_generate(model, prompts, batch_size, max_gen_len, options, disable_cache=False):
    """The loop of the reference's tests: prompts taken `batch_size` at a time, padded, generated; best hypotheses in
    prompt order."""
    from lcm.datasets.batch import EmbeddingsBatch, pad_seqs
    from lcm.inference.lcm import LCMGenerator, LCMGeneratorOptions
    gen = LCMGenerator(model, eos_vec=torch.zeros(prompts[0].shape[1], device="cuda:0"), options=LCMGeneratorOptions(**options))
    outs = []
    for i in range(0, len(prompts), batch_size):
        padded = EmbeddingsBatch(*pad_seqs([p.cuda() for p in prompts[i:i + batch_size]]))
        outs.extend(h[0].seq for h in gen(padded, max_gen_len=max_gen_len, disable_cache=disable_cache).hypotheses)
    return outs


@pytest.mark.parametrize("disable_cache,eos_threshold", [(True, 0.3), (True, 1.1), (False, 0.3), (False, 1.1)])
def test_reference_caching_and_batching(disable_cache, eos_threshold):
    """The reference's tests/units/inference/test_base_lcm_batched_inference.py:21-96 at a width these kernels take:
    five prompts of random lengths generated three at a time and one at a time, with and without the KV cache, a zero
    EOS vector under both of its thresholds, no sampling: the outputs agree to atol 1e-6."""
    from lcm.models.base_lcm import create_base_lcm_model
    from oracle.weights import synthetic_prompts
    g = torch.Generator().manual_seed(17)
    model = create_base_lcm_model(base_config("mini_base_lcm"), device="cuda:0", dtype=torch.float32, seed=3)
    full = synthetic_prompts(oracle.load_dummy_normalizer(), 5, 7, seed=31)
    prompts = [full[i, :int(torch.randint(1, 8, (1,), generator=g))] for i in range(5)]
    opts = dict(eos_threshold=eos_threshold, sample_latent_variable=False)
    batched = _reference_style_generate(model, prompts, 3, 8, opts, disable_cache)
    single = _reference_style_generate(model, prompts, 1, 8, opts, disable_cache)
    assert len(batched) == len(single) == 5
    for a, b, p in zip(batched, single, prompts):
        assert a.shape == b.shape and a.shape[0] == p.shape[0] + 8 and torch.isfinite(a).all()
        assert torch.allclose(a, b, atol=1e-6), float((a - b).abs().max())


@pytest.mark.parametrize("batch_size", [3, 1])
def test_reference_single_input_stopping(batch_size):
    """tests/units/inference/test_base_lcm_batched_inference.py:99-155: with an EOS criterion that never fires, no
    document stops before `max_gen_len`, in a batch of three and alone."""
    from lcm.models.base_lcm import create_base_lcm_model
    from oracle.weights import synthetic_prompts
    g = torch.Generator().manual_seed(batch_size)
    model = create_base_lcm_model(base_config("mini_base_lcm"), device="cuda:0", dtype=torch.float32, seed=3)
    full = synthetic_prompts(oracle.load_dummy_normalizer(), batch_size, 4, seed=32)
    prompts = [full[i, :int(torch.randint(1, 5, (1,), generator=g))] for i in range(batch_size)]
    outs = _reference_style_generate(model, prompts, 3, 3, dict(eos_threshold=1, sample_latent_variable=False, trim_hypotheses=False))
    for o, p in zip(outs, prompts):
        assert o.shape[0] - p.shape[0] == 3


@pytest.mark.parametrize("prefix_len", [1, 2, 8])
def test_reference_kv_caching(prefix_len):
    """tests/units/inference/test_base_lcm_kv_caching.py:17-74 at a width these kernels take: one document of 1 / 2 / 8
    prompt sentences (with one sentence the prefill is a no-op), 8 generated, stopping rules that never fire, no
    sampling, a seed; without and with the KV cache the outputs agree to atol 1e-3."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.inference.lcm import LCMGenerator, LCMGeneratorOptions
    from lcm.models.base_lcm import create_base_lcm_model
    from oracle.weights import synthetic_prompts
    model = create_base_lcm_model(base_config("mini_base_lcm"), device="cuda:0", dtype=torch.float32, seed=3)
    x = synthetic_prompts(oracle.load_dummy_normalizer(), 1, prefix_len, seed=40 + prefix_len).cuda()
    gen = LCMGenerator(model, eos_vec=torch.zeros(1024, device="cuda:0"),
                       options=LCMGeneratorOptions(eos_threshold=1, stop_on_repetition_cosine_threshold=1,
                                                   sample_latent_variable=False, seed=0))
    no_cache = gen(EmbeddingsBatch(seqs=x, padding_mask=None), max_gen_len=8, disable_cache=True)
    cached = gen(EmbeddingsBatch(seqs=x, padding_mask=None), max_gen_len=8, disable_cache=False)
    a, b = no_cache.hypotheses[0][0].seq, cached.hypotheses[0][0].seq
    assert a.shape == b.shape == (prefix_len + 8, 1024) and torch.isfinite(a).all()
    assert torch.allclose(a, b, atol=1e-3), float((a - b).abs().max())


def test_reference_next_sentence_prediction():
    """tests/units/models/test_lcm_architecture.py:78-112, call for call: the toy architecture taken from the model-type
    registry's config loader, built by its model factory; `predict_next_sentence` over all positions is deterministic
    without sampling and random with it."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.base_lcm import BASE_LCM_MODEL_TYPE
    from lcm.utils.model_type_registry import lcm_model_type_registry
    model_type_cfg = lcm_model_type_registry.get_config(BASE_LCM_MODEL_TYPE)
    model_cfg = model_type_cfg.config_loader._arch_configs.get(f"toy_{BASE_LCM_MODEL_TYPE}")
    model = model_type_cfg.model_factory(model_cfg)
    model.eval()
    x = (torch.randn(size=[5, 17, 1024]) * 0.006).cuda()
    batch = EmbeddingsBatch(x, padding_mask=None)
    first = model.predict_next_sentence(batch, sample=False).seqs
    assert first.shape == x.shape and torch.isfinite(first).all()
    assert torch.allclose(first, model.predict_next_sentence(batch, sample=False).seqs)
    random_prediction = model.predict_next_sentence(batch, sample=True, temperature=0.01).seqs
    assert random_prediction.shape == x.shape
    assert not torch.allclose(random_prediction, first)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_forward_stays_in_normalizer_space(dtype):
    """`BaseLCModel.forward` has no normalizer on either side (reference builder.py:94-124); `predict_next_sentence`
    wraps it in normalize / denormalize (:150-165).  So predict(raw) == denormalize(forward(normalize(raw))) with the
    model's own normalizer, and forward against the oracle's forward directly."""
    from lcm.datasets.batch import EmbeddingsBatch
    cfg, sd, x, _ = make_base_problem("mini_base_lcm", 3, 6, 1, seed=9, norm_jitter=0.2)
    from lcm.models.base_lcm import create_base_lcm_model
    model = create_base_lcm_model(base_config("mini_base_lcm"), device="cuda:0", dtype=dtype)
    model.load_state_dict(sd)
    normalizer = model.frontend.sonar_normalizer
    raw = EmbeddingsBatch(x.cuda().to(dtype), None)
    predicted = model.predict_next_sentence(raw).seqs.float()
    inner = model.forward(raw.normalize_seqs(normalizer))
    by_hand = inner.denormalize_seqs(normalizer).seqs.float()
    scale = sd["frontend.sonar_normalizer.scale"].cuda()
    center = sd["frontend.sonar_normalizer.center"].cuda()
    if dtype == torch.float32:
        assert float(((predicted - by_hand) / scale).abs().max()) <= 1e-5
    else:       # the stand-alone normalizer hands bf16 rows to forward, the fused path rounds once: two bf16 evaluations
        a, b = (predicted - center) / scale, (by_hand - center) / scale
        assert float(torch.nn.functional.cosine_similarity(a, b, dim=-1).min()) >= BF16_COS
    om = oracle.OracleBaseLCM(cfg, {k: (oracle.r16(v) if dtype == torch.bfloat16 else v) for k, v in sd.items()})
    xn = om.normalize(x)
    want = om.forward(oracle.r16(xn) if dtype == torch.bfloat16 else xn)
    got = inner.seqs.float().cpu()
    cos = float(torch.nn.functional.cosine_similarity(got, want, dim=-1).min())
    if dtype == torch.float32:
        assert float((got - want).abs().max()) <= FP32_MAX_ABS and cos >= FP32_COS, (float((got - want).abs().max()), cos)
    else:
        assert cos >= BF16_COS, cos
