"""Checks that keep the self-made CPU oracle honest (no GPU needed).

The reference ships no numeric golden vectors for the two-tower generation path and cannot be
imported here, so the oracle is pinned by: constants the reference's own files state
(parameter counts, timestep lists), table anchors, two independent statements of RoPE / SDPA,
the reference's *invariant* tests restated for this path (KV cache on/off, batched vs single,
determinism -- tests/units/inference/test_base_lcm_kv_caching.py:17-74,
test_base_lcm_batched_inference.py:21-155, tests/units/models/test_lcm_architecture.py:78-112)
and analytic identities of the algorithm.
"""

import math
import os

import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle.archs import SchedulerConfig
from oracle.model import KVCache, causal_mask, linear, rope, rope_explicit, sdpa, sinusoidal_positions
from oracle.weights import num_parameters, synthetic_noise, synthetic_prompts

OPTS = dict(guidance_scale=3.0, guidance_rescale=0.7, initial_noise_scale=0.6, eos_threshold=None)


def _model(arch, seed=0, **kw):
    cfg = oracle.get_arch(arch)
    init = dict(mod_gain=10.0, qk_gain=2.0)
    init.update(kw)
    return cfg, oracle.OracleTwoTower(cfg, oracle.init_state_dict(cfg, seed=seed, **init))


# ----------------------------------------------------------------------------- constants of the reference
def test_parameter_counts_match_reference_docstrings():
    # lcm/models/two_tower_diffusion_lcm/archs.py:73-74 states 6,930,781,696 for the 7B arch: the
    # state-dict shapes (GLU rounding to 5632 / 11008, biases, modulators) reproduce it exactly.
    assert num_parameters(oracle.get_arch("two_tower_diffusion_lcm_7B")) == 6_930_781_696
    # The 1.6B docstring (archs.py:29-30: 1,635,101,696) differs from the count its own config gives.
    assert num_parameters(oracle.get_arch("two_tower_diffusion_lcm_1_6B")) == 1_645_538_304
    from oracle.archs import glu_inner_dim
    assert glu_inner_dim(8192) == 5632 and glu_inner_dim(4096) == 2816 and glu_inner_dim(16384) == 11008


def test_timestep_lists():
    # lcm/nn/schedulers/ddim.py:333-347: the integer arange truncates the step (T=100, S=40 -> 50 steps)
    s = oracle.DDIMSchedule(SchedulerConfig(num_train_steps=100))
    s.set_timesteps(40)
    assert s.timesteps.tolist() == list(range(99, 0, -2))
    s.set_timesteps(100)
    assert s.timesteps.tolist() == list(range(99, -1, -1))
    s = oracle.DDIMSchedule(SchedulerConfig(num_train_steps=1000))
    s.set_timesteps(10)
    assert s.timesteps.tolist() == [999, 899, 799, 699, 599, 499, 399, 299, 199, 99]   # ddim.py:335 comment
    with pytest.raises(ValueError):
        s.set_timesteps(1001)


def test_alpha_table_anchors():
    s100 = oracle.DDIMSchedule(SchedulerConfig(num_train_steps=100))
    ac = s100.alphas_cumprod
    assert abs(float(ac[0]) - 0.99936867) < 1e-6 and abs(float(ac[50]) - 0.47805437) < 1e-6
    assert abs(float(ac[98]) - 2.2796521e-4) < 1e-8 and float(ac[99]) == 0.0       # zero terminal SNR
    s1000 = oracle.DDIMSchedule(SchedulerConfig(num_train_steps=1000))
    assert abs(float(s1000.alphas_cumprod[500]) - 0.49226481) < 1e-6 and float(s1000.alphas_cumprod[999]) == 0.0
    assert torch.all(ac[1:] <= ac[:-1])


# ----------------------------------------------------------------------------- primitives, stated twice
def test_rope_two_statements_agree():
    x = torch.randn(2, 3, 5, 128)
    # the complex form rounds the angle to fp32 (like fairseq2); the explicit form uses double angles, so the
    # tolerance grows with the position (angle error ~ pos * 6e-8 rad)
    for start, tol in ((0, 2e-5), (7, 2e-5), (1000, 5e-4)):
        assert torch.allclose(rope(x, start, 10000.0), rope_explicit(x, start, 10000.0), atol=tol)
    assert torch.allclose(rope(x, 0, 10000.0)[..., 0, :], x[..., 0, :], atol=1e-6)      # position 0 is the identity
    # rotating pairs preserves the norm of every pair
    y = rope(x, 13, 10000.0)
    assert torch.allclose(y.unflatten(-1, (-1, 2)).norm(dim=-1), x.unflatten(-1, (-1, 2)).norm(dim=-1), atol=1e-5)


def test_sdpa_matches_torch():
    q, k, v = torch.randn(3, 4, 5, 128), torch.randn(3, 4, 9, 128), torch.randn(3, 4, 9, 128)
    assert torch.allclose(sdpa(q, k, v, None, None), F.scaled_dot_product_attention(q, k, v), atol=1e-5)
    lens = torch.tensor([9, 4, 1])
    keep = (torch.arange(9)[None] < lens[:, None])[:, None, None, :]
    assert torch.allclose(sdpa(q, k, v, lens, None), F.scaled_dot_product_attention(q, k, v, attn_mask=keep), atol=1e-5)
    m = causal_mask(5, 9)
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=(m == 0))
    assert torch.allclose(sdpa(q, k, v, None, m), ref, atol=1e-5)
    assert causal_mask(1, 9) is None and causal_mask(4, 4).shape == (4, 4)


# ----------------------------------------------------------------------------- invariants of the reference tests
@pytest.mark.parametrize("arch", ["toy_two_tower_diffusion_lcm", "mini_two_tower_diffusion_lcm"])
@pytest.mark.parametrize("prefix", [1, 2, 8])
def test_kv_cache_equals_full_encode(arch, prefix):
    cfg, m = _model(arch)
    x = torch.randn(2, prefix + 3, 1024)
    full = m.encode(x)
    cache = KVCache(32)
    parts = [m.encode(x[:, :prefix], cache)]
    cache.increment_step_nr(prefix)
    for t in range(prefix, prefix + 3):
        parts.append(m.encode(x[:, t:t + 1], cache))
        cache.increment_step_nr(1)
    assert torch.allclose(torch.cat(parts, 1), full, atol=1e-3)        # tolerance of the reference test


@pytest.mark.parametrize("arch", ["toy_two_tower_diffusion_lcm", "mini_two_tower_diffusion_lcm"])
def test_batched_equals_single_and_is_deterministic(arch):
    cfg, m = _model(arch)
    norm = oracle.load_dummy_normalizer()
    lens = [2, 4, 3]
    x = synthetic_prompts(norm, 3, 4)
    for i, n in enumerate(lens):
        x[i, n:] = 0
    nz = synthetic_noise(8, 3)
    gen = oracle.OracleGenerator(m, oracle.GenOptions(**OPTS, inference_timesteps=5))
    out = gen(x, prompt_lens=torch.tensor(lens), max_gen_len=3, noise=nz)
    again = gen(x, prompt_lens=torch.tensor(lens), max_gen_len=3, noise=nz)
    for i, n in enumerate(lens):
        assert out[i][0].seq.shape[0] == n + 3                       # never stops before max_gen_len
        assert torch.equal(out[i][0].seq, again[i][0].seq)
        # noise is consumed as noise[step - min_prompt_len, doc]: a single document starts later
        single = gen(x[i:i + 1, :n], max_gen_len=3, noise=nz[n - min(lens):, i:i + 1])
        assert torch.allclose(single[0][0].seq, out[i][0].seq, atol=1e-5), i


def test_guidance_scale_one_plus_eps_approaches_unguided_path():
    cfg, m = _model("mini_two_tower_diffusion_lcm")
    x, nz = synthetic_prompts(oracle.load_dummy_normalizer(), 2, 3), synthetic_noise(2, 2)
    base = dict(OPTS, inference_timesteps=10, guidance_rescale=0.0)
    plain = oracle.OracleGenerator(m, oracle.GenOptions(**dict(base, guidance_scale=1.0)))(x, max_gen_len=2, noise=nz)
    cfgd = oracle.OracleGenerator(m, oracle.GenOptions(**dict(base, guidance_scale=1.0 + 1e-6)))(x, max_gen_len=2, noise=nz)
    c, s = m.sd["sonar_normalizer.center"], m.sd["sonar_normalizer.scale"]
    for a, b in zip(plain, cfgd):
        assert torch.allclose((a[0].seq - c) / s, (b[0].seq - c) / s, atol=2e-3)


# ----------------------------------------------------------------------------- analytic identities
def test_zero_modulators_make_denoiser_layers_identity():
    cfg = oracle.get_arch("mini_two_tower_diffusion_lcm")
    sd = oracle.init_state_dict(cfg, seed=3, mod_gain=0.0)           # the reference's zero-init (lcm_denoiser.py:55-58)
    m = oracle.OracleTwoTower(cfg, sd)
    lat, ctx = torch.randn(2, 1, 1024), torch.randn(2, 4, cfg.model_dim)
    out = m.denoise(lat, torch.full((2, 1), 50), ctx)
    from oracle.model import linear, norm
    ref = linear(norm(linear(lat, sd, "denoiser.initial_proj.fc"), sd, "denoiser.layer_norm", "rms", cfg.norm_eps), sd,
                 "denoiser.final_proj.fc")
    assert torch.allclose(out, ref, atol=1e-5)


def test_unconditional_cross_attention_is_a_constant():
    """Rows with context length 0 only see the prepended zero vector: output_proj(v_proj(0)) = W_o b_v + b_o,
    whatever the (masked) context holds."""
    cfg, m = _model("mini_two_tower_diffusion_lcm")
    lat = torch.randn(2, 1, 1024)
    ts = torch.full((2, 1), 30)
    a = m.denoise(lat, ts, torch.randn(2, 5, cfg.model_dim), torch.zeros(2, dtype=torch.long))
    b = m.denoise(lat, ts, torch.randn(2, 5, cfg.model_dim), torch.zeros(2, dtype=torch.long))
    assert torch.allclose(a, b, atol=1e-6)
    sd, p = m.sd, "denoiser.layers.0.cross_attention"
    h = torch.randn(1, 1, cfg.model_dim)
    got = m._mha(p, cfg.denoiser, h, torch.zeros(1, 1, cfg.model_dim), key_lens=torch.tensor([1]))
    want = sd[p + ".output_proj.weight"] @ sd[p + ".v_proj.bias"] + sd[p + ".output_proj.bias"]
    assert torch.allclose(got.flatten(), want, atol=1e-5)


def test_ddim_end_points():
    s = oracle.DDIMSchedule(SchedulerConfig(num_train_steps=100))
    s.set_timesteps(40)
    x, out = torch.randn(3, 1, 1024), torch.randn(3, 1, 1024)
    # first step, alpha_bar = 0: "sample" prediction gives x0 = model output and eps = x_t
    prev, x0 = s.step(out, 99, x)
    assert torch.equal(x0, out)
    a_prev = s.alphas_cumprod[97]
    assert torch.allclose(prev, a_prev**0.5 * out + (1 - a_prev) ** 0.5 * x, atol=1e-6)
    # last step: previous alpha_bar = 1 -> prev_sample == pred_original_sample
    prev, x0 = s.step(out, 1, x)
    assert torch.allclose(prev, x0, atol=1e-6)
    # get_epsilon inverts the "sample" parameterisation
    eps = s.get_epsilon(out, x, 57)
    a = s.alphas_cumprod[57]
    assert torch.allclose(a**0.5 * out + (1 - a) ** 0.5 * eps, x, atol=1e-5)


def test_cfg_rescale_properties():
    class M:
        guidance_scale, guidance_rescale = 3.0, 1.0
    eps = torch.randn(8, 1, 1024)
    g = oracle.OracleTwoTower.apply_cfg(M, eps)
    # full rescale: the guided noise has the std of the conditional prediction
    assert torch.allclose(g.std(-1), eps.chunk(2)[0].std(-1), rtol=1e-5)
    M.guidance_rescale, M.guidance_scale = 0.0, 1.0
    assert torch.allclose(oracle.OracleTwoTower.apply_cfg(M, eps), eps.chunk(2)[0], atol=1e-6)


# ----------------------------------------------------------------------------- regression fixture
def test_golden_fixture_regression():
    blob = torch.load(os.path.join(os.path.dirname(__file__), "golden", "toy_two_tower_golden.pt"), weights_only=True)
    cfg = oracle.get_arch(blob["arch"])
    sd = oracle.init_state_dict(cfg, **blob["init"])
    trace = []
    gen = oracle.OracleGenerator(oracle.OracleTwoTower(cfg, sd), oracle.GenOptions(**blob["options"]))
    hyps = gen(blob["prompts"], max_gen_len=blob["G"], noise=blob["noise"], trace=trace)
    assert torch.equal(gen.model.scheduler.timesteps, blob["timesteps"])
    out = torch.stack([h[0].seq for h in hyps])
    scale = sd["sonar_normalizer.scale"]
    assert ((out - blob["outputs"]) / scale).abs().max() <= 1e-4          # normalizer units
    assert (trace[0][1] - blob["latents_step0"]).abs().max() <= 1e-5
    assert (trace[len(blob["timesteps"]) - 1][2] - blob["x0_last"]).abs().max() <= 1e-4


def test_option_statements_fixture_regression():
    """tests/golden/option_statements_golden.pt (oracle/make_golden_options.py): the oracle's statements of the layer
    orders, residual scale, sinusoidal positions, PReLU, postnet activation and packed FFT still give the committed
    vectors."""
    from oracle import make_golden_options as G
    from oracle.model import fft_inverse_transform, fft_transform, sinusoidal_positions
    blob = torch.load(os.path.join(os.path.dirname(__file__), "golden", "option_statements_golden.pt"), weights_only=True)
    assert set(blob["base"]) == set(G.BASE_CASES)
    for name, case in blob["base"].items():
        cfg, sd, x = G.base_case(name)
        assert torch.equal(x, case["prompts"]), name
        out = oracle.OracleBaseLCM(cfg, sd).predict(x)
        assert (out - case["predict"]).abs().max() <= 1e-5 * case["predict"].abs().max(), name
    f = blob["fft"]
    assert (fft_transform(f["x"]) - f["packed"]).abs().max() <= 1e-5 * f["packed"].abs().max()
    assert (fft_inverse_transform(f["packed"]) - f["back"]).abs().max() <= 1e-5
    assert (f["back"] - f["x"]).abs().max() <= 1e-4
    assert (sinusoidal_positions(3, 5, 256) - blob["sine_rows_3_to_7_dim_256"]).abs().max() <= 1e-6


def test_dummy_normalizer_fixture():
    n = oracle.load_dummy_normalizer()
    assert n["center"].shape == (1024,) and n["scale"].shape == (1024,) and n["center"].dtype == torch.float32
    assert float(n["scale"].min()) > 0
    x = torch.randn(4, 1024)
    cfg, m = _model("toy_two_tower_diffusion_lcm")
    assert torch.allclose(m.denormalize(m.normalize(x)), x, atol=1e-5)


# ---- base LCM oracle ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("arch", ["toy_base_lcm", "mini_base_lcm"])
def test_base_lcm_cache_equals_full_forward(arch):
    """tests/units/inference/test_base_lcm_kv_caching.py of the reference, on the oracle: prefix-by-prefix decoding
    through the KV cache equals one full causal forward."""
    from oracle.model import KVCache
    cfg = oracle.get_base_arch(arch)
    sd = oracle.init_state_dict(cfg, seed=3, qk_gain=2.0, pos_std=0.5)
    m = oracle.OracleBaseLCM(cfg, sd)
    x = synthetic_prompts(oracle.load_dummy_normalizer(), 2, 6, seed=5) * (1.0 if m.has_normalizer else 50.0)
    full = m.predict(x)
    cache = KVCache(6)
    parts = []
    for a, b in [(0, 3), (3, 4), (4, 6)]:
        parts.append(m.predict(x[:, a:b], cache))
        cache.increment_step_nr(b - a)
    got = torch.cat(parts, 1)
    assert (got - full).abs().max() <= 1e-5 * full.abs().max()


def test_base_lcm_generator_batched_equals_single_and_sampling():
    """tests/units/inference/test_base_lcm_batched_inference.py of the reference, on the oracle; plus: temperature
    noise is added in normalized space (base_lcm/builder.py:158-165)."""
    cfg = oracle.get_base_arch("mini_base_lcm")
    sd = oracle.init_state_dict(cfg, seed=4, qk_gain=2.0)
    m = oracle.OracleBaseLCM(cfg, sd)
    x = synthetic_prompts(oracle.load_dummy_normalizer(), 3, 3, seed=6)
    nz = synthetic_noise(3, 3, seed=7)
    gen = oracle.OracleGenerator(m, oracle.GenOptions(eos_threshold=None, lcm_temperature=0.5))
    full = gen(x, max_gen_len=3, noise=nz)
    for i in range(3):
        one = gen(x[i:i + 1], max_gen_len=3, noise=nz[:, i:i + 1])
        assert torch.allclose(one[0][0].seq, full[i][0].seq, atol=1e-6 * float(full[i][0].seq.abs().max()) + 1e-9)
    det = oracle.OracleGenerator(m, oracle.GenOptions(eos_threshold=None, sample_latent_variable=False))(x, max_gen_len=1, noise=nz)
    smp = oracle.OracleGenerator(m, oracle.GenOptions(eos_threshold=None, lcm_temperature=0.5))(x, max_gen_len=1, noise=nz)
    s = sd["frontend.sonar_normalizer.scale"]
    delta = (smp[0][0].seq[-1] - det[0][0].seq[-1]) / s
    assert torch.allclose(delta, 0.5 * nz[0, 0], atol=1e-4)


def test_base_lcm_parameter_count():
    from oracle.weights import base_state_dict_spec
    import math
    n = sum(math.prod(s) for name, s, kind in base_state_dict_spec(oracle.get_base_arch("base_lcm_1_6B")) if kind != "normalizer")
    assert n == 1_649_185_792      # the reconstructed 1.6B decoder; the reference's recipe comment states 1,647,635,456


def test_scorer_prediction_equals_first_generated_sentence():
    """Teacher forcing: the scorer's prediction for position t of a document is what the generator would produce as
    its first sentence when prompted with the gold prefix [:t] and the same noise draw."""
    cfg, m = _model("mini_two_tower_diffusion_lcm")
    norm = oracle.load_dummy_normalizer()
    lens = [5, 3, 4]
    x = synthetic_prompts(norm, 3, 5)
    for i, n in enumerate(lens):
        x[i, n:] = 0
    nz = synthetic_noise(4, 3)
    opts = oracle.GenOptions(**OPTS, inference_timesteps=5)
    for mcl in (1, 2):
        scored = oracle.OracleScorer(m, opts)(x, text_lens=torch.tensor(lens), min_context_len=mcl, noise=nz)
        gen = oracle.OracleGenerator(m, opts)
        for i, n in enumerate(lens):
            assert scored[i][0].seq.shape == (n - mcl, x.shape[-1])
            for t in range(mcl, n):
                first = gen(x[i:i + 1, :t], max_gen_len=1, noise=nz[t - mcl:t - mcl + 1, i:i + 1])[0][0].seq[-1]
                assert torch.allclose(scored[i][0].seq[t - mcl], first, atol=1e-5), (mcl, i, t)


def test_training_shaped_forward_equals_per_position_inference_denoise():
    """The all-positions forward under the shifted causal mask gives, at position j, what the inference-path denoiser
    gives for that latent with the first j context positions (two statements of the same masking)."""
    cfg, m = _model("mini_two_tower_diffusion_lcm")
    norm = oracle.load_dummy_normalizer()
    x = m.normalize(synthetic_prompts(norm, 2, 5))
    g = torch.Generator().manual_seed(5)
    noisy = torch.randn(2, 5, x.shape[-1], generator=g)
    ts = torch.randint(0, 100, (2, 5), generator=g)
    full = m.forward(x, noisy, ts)
    ctx = m.encode(x)
    for j in range(5):
        # position 0 sees only the zero vector: stated like the unconditional rows of guidance (no visible context)
        one = (m.denoise(noisy[:, j:j + 1], ts[:, j:j + 1], ctx[:, :j]) if j > 0 else
               m.denoise(noisy[:, :1], ts[:, :1], ctx[:, :1], ctx_lens=torch.zeros(2, dtype=torch.long)))
        assert torch.allclose(full[:, j:j + 1], one, atol=1e-4), j
    # source_lengths move the visible window: with shift s position j sees the first j + s context positions
    shifted = m.forward(x, noisy, ts, source_lengths=torch.tensor([2, 1]))
    for b, s in enumerate([2, 1]):
        for j in range(5):
            one = m.denoise(noisy[b:b + 1, j:j + 1], ts[b:b + 1, j:j + 1], ctx[b:b + 1, :min(5, j + s)])
            assert torch.allclose(shifted[b:b + 1, j:j + 1], one, atol=1e-4), (b, j)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_norm_styles_against_torch_modules(dtype):
    """Independent statements of lcm/nn/normalization.py:16-88 through torch's own modules: `rms` against
    torch.nn.functional.rms_norm (fp32 statistics, cast back, then the weight: fairseq2's RMSNorm order), `standard`
    against torch.nn.LayerNorm with weight and bias, `fp32` against LayerNorm evaluated on fp32 copies and cast back
    (:17-33), `unit` against LayerNorm(elementwise_affine=False) (:56-70)."""
    from oracle.model import norm
    torch.manual_seed(3)
    d, eps = 256, 1e-5
    x = (torch.randn(7, 5, d) * 3.0 + 0.5).to(dtype)
    w, b = (1.0 + 0.3 * torch.randn(d)).to(dtype), (0.2 * torch.randn(d)).to(dtype)
    sd = {"n.weight": w, "n.bias": b}
    # rms: normalise in fp32 without a weight, cast to the input dtype, multiply by the weight in that dtype
    want = F.rms_norm(x.float(), (d,), None, eps).to(dtype) * w
    assert torch.equal(norm(x, sd, "n", "rms", eps), want)
    ln = torch.nn.LayerNorm(d, eps=eps, dtype=dtype)
    with torch.no_grad():
        ln.weight.copy_(w)
        ln.bias.copy_(b)
    assert torch.equal(norm(x, sd, "n", "standard", eps), ln(x))
    ln32 = torch.nn.LayerNorm(d, eps=eps, bias=False)
    with torch.no_grad():
        ln32.weight.copy_(w.float())
    assert torch.equal(norm(x, sd, "n", "fp32", eps), ln32(x.float()).to(dtype))
    unit = torch.nn.LayerNorm(d, eps=eps, elementwise_affine=False)
    assert torch.equal(norm(x, {}, "n", "unit", eps), unit(x))
    # the definitions themselves, in float64
    x64 = x.double()
    rms64 = x64 / (x64.pow(2).mean(-1, keepdim=True) + eps).sqrt() * w.double()
    tol = 1e-5 if dtype == torch.float32 else 4e-2
    assert (norm(x, sd, "n", "rms", eps).double() - rms64).abs().max() < tol * rms64.abs().max()
    mu, var = x64.mean(-1, keepdim=True), x64.var(-1, unbiased=False, keepdim=True)
    ln64 = (x64 - mu) / (var + eps).sqrt()
    assert (norm(x, {}, "n", "unit", eps).double() - ln64).abs().max() < tol * ln64.abs().max()
    assert (norm(x, sd, "n", "standard", eps).double() - (ln64 * w.double() + b.double())).abs().max() < tol * 8


def test_base_scorer_equals_one_causal_forward():
    """The base LCM's teacher-forced scorer (lcm/inference/lcm/scorer.py:19-198: one cached decode step per sentence,
    gold inputs) is one causal forward over the gold sequence: prediction j of the scorer == position j - 1 of
    `predict` over the whole document, for ragged documents and min_context_len 1 and 3."""
    cfg = oracle.get_base_arch("mini_base_lcm")
    sd = oracle.init_state_dict(cfg, seed=3, qk_gain=2.0)
    lens = [6, 4, 5]
    x = synthetic_prompts(oracle.load_dummy_normalizer(), 3, 6, seed=9)
    for i, n in enumerate(lens):
        x[i, n:] = 0
    model = oracle.OracleBaseLCM(cfg, sd)
    full = model.predict(x)                                                  # (3, 6, D): all positions, no cache
    for mcl in (1, 3):
        hyps = oracle.OracleScorer(model, oracle.GenOptions(eos_threshold=None, sample_latent_variable=False))(
            x, text_lens=torch.tensor(lens), min_context_len=mcl)
        for i, n in enumerate(lens):
            got = hyps[i][0].seq
            assert got.shape == (n - mcl, 1024)
            assert torch.allclose(got, full[i, mcl - 1:n - 1], atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("order,final,pos,swiglu", [("post", None, "sine", True), ("normformer", None, "rope", False),
                                                   ("pre", "post", "sine", True), ("post", "pre", "learned", False)])
def test_contextualizer_options_cached_equals_full(order, final, pos, swiglu):
    """The reference's KV-caching invariant (tests/units/inference/test_base_lcm_kv_caching.py:17-74) on the oracle's
    statements of the non-default layer options: encoding position by position through the cache equals one causal
    pass, so the order of norms / residual scale / sinusoidal offset is applied per position and not per call."""
    cfg = oracle.get_arch("mini_two_tower_diffusion_lcm")
    cx = cfg.context
    cx.norm_order, cx.final_norm_order, cx.pos, cx.use_swiglu, cx.scale_residual = order, final, pos, swiglu, True
    cx.norm = "standard" if not swiglu else "rms"
    sd = oracle.init_state_dict(cfg, seed=2, qk_gain=1.5, norm_jitter=0.2)
    model = oracle.OracleTwoTower(cfg, sd)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 7, 1024, generator=g)
    full = model.encode(x)
    cache = KVCache(16)
    steps = []
    for j in range(7):
        steps.append(model.encode(x[:, j:j + 1], cache))
        cache.increment_step_nr(1)
    assert torch.allclose(torch.cat(steps, dim=1), full, atol=2e-5, rtol=1e-4)
    # and the sinusoidal table is the closed form
    tab = sinusoidal_positions(3, 2, 8)
    f = torch.exp(torch.arange(4, dtype=torch.float64) * (-math.log(10000.0) / 3))
    want = torch.cat([torch.sin(torch.tensor([[3.0], [4.0]], dtype=torch.float64) * f), torch.cos(torch.tensor([[3.0], [4.0]], dtype=torch.float64) * f)], -1)
    assert torch.allclose(tab.double(), want, atol=1e-6)


def test_decoder_stack_against_an_independent_llama_implementation():
    """The oracle's pre-norm RMSNorm / RoPE / SwiGLU decoder stack (the base LCM's `lcm` tower: frontend projection,
    three layers, final norm) against Hugging Face `transformers`' LlamaModel, an implementation written by other
    people, fed the same weights.  The one convention that differs is the rotary pairing: the oracle follows
    fairseq2 / the original LLaMA code (consecutive pairs (2i, 2i + 1)), transformers rotates half against half, and
    the two are equal once the rows of W_q / W_k (and their biases) are permuted even-indices-first inside every head
    -- the permutation transformers' own LLaMA checkpoint converter applies.  Everything else (RMSNorm cast order,
    causal mask, softmax scale, SwiGLU gate order, residuals) must agree as it stands."""
    transformers = pytest.importorskip("transformers")
    cfg = oracle.get_base_arch("mini_base_lcm")
    tc = cfg.lcm
    sd = oracle.init_state_dict(cfg, seed=6, qk_gain=2.0, norm_jitter=0.1, b_std=0.05)
    inner = sd["lcm.layers.0.ffn.inner_proj.weight"].shape[0]
    hf_cfg = transformers.LlamaConfig(
        vocab_size=8, hidden_size=cfg.model_dim, intermediate_size=inner, num_hidden_layers=tc.num_layers,
        num_attention_heads=tc.num_heads, num_key_value_heads=tc.num_heads, rms_norm_eps=cfg.norm_eps,
        rope_theta=tc.rope_theta, attention_bias=True, mlp_bias=True, max_position_embeddings=64,
        attn_implementation="eager")
    hf = transformers.LlamaModel(hf_cfg).eval()

    def halves_first(w):        # rows (head, pair, 2) -> (head, 2, pair)
        hd = cfg.model_dim // tc.num_heads
        return w.reshape(tc.num_heads, hd // 2, 2, *w.shape[1:]).transpose(1, 2).reshape(w.shape)

    with torch.no_grad():
        for i, layer in enumerate(hf.layers):
            p = f"lcm.layers.{i}"
            a = layer.self_attn
            a.q_proj.weight.copy_(halves_first(sd[p + ".self_attn.q_proj.weight"]))
            a.q_proj.bias.copy_(halves_first(sd[p + ".self_attn.q_proj.bias"]))
            a.k_proj.weight.copy_(halves_first(sd[p + ".self_attn.k_proj.weight"]))
            a.k_proj.bias.copy_(halves_first(sd[p + ".self_attn.k_proj.bias"]))
            a.v_proj.weight.copy_(sd[p + ".self_attn.v_proj.weight"])
            a.v_proj.bias.copy_(sd[p + ".self_attn.v_proj.bias"])
            a.o_proj.weight.copy_(sd[p + ".self_attn.output_proj.weight"])
            a.o_proj.bias.copy_(sd[p + ".self_attn.output_proj.bias"])
            layer.input_layernorm.weight.copy_(sd[p + ".self_attn_layer_norm.weight"])
            layer.post_attention_layernorm.weight.copy_(sd[p + ".ffn_layer_norm.weight"])
            layer.mlp.gate_proj.weight.copy_(sd[p + ".ffn.gate_proj.weight"])
            layer.mlp.gate_proj.bias.copy_(sd[p + ".ffn.gate_proj.bias"])
            layer.mlp.up_proj.weight.copy_(sd[p + ".ffn.inner_proj.weight"])
            layer.mlp.up_proj.bias.copy_(sd[p + ".ffn.inner_proj.bias"])
            layer.mlp.down_proj.weight.copy_(sd[p + ".ffn.output_proj.weight"])
            layer.mlp.down_proj.bias.copy_(sd[p + ".ffn.output_proj.bias"])
        hf.norm.weight.copy_(sd["lcm.layer_norm.weight"])

    model = oracle.OracleBaseLCM(cfg, sd)
    torch.manual_seed(2)
    x_norm = torch.randn(3, 11, cfg.sonar_dim)
    with torch.no_grad():
        want = model.forward(x_norm)                                           # frontend + stack + final norm + postnet
        hidden = hf(inputs_embeds=linear(x_norm, sd, "frontend.pre_linear")).last_hidden_state
        got = linear(hidden, sd, "postnet.fc")
    assert got.shape == want.shape
    assert float((got - want).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max())), float((got - want).abs().max())
    # and position by position through the oracle's KV cache (what the generators use)
    cache = KVCache(16)
    with torch.no_grad():
        steps = []
        for j in range(x_norm.shape[1]):
            steps.append(model.forward(x_norm[:, j:j + 1], cache))
            cache.increment_step_nr(1)
    assert float((torch.cat(steps, dim=1) - got).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
