"""Shared helpers of the parity tests: run the same seeded problem through the CPU oracle and
through the HIP-backed `lcm` package."""

import torch

import oracle
from oracle.weights import synthetic_noise, synthetic_prompts

DEFAULT_OPTS = dict(guidance_scale=3.0, guidance_rescale=0.7, initial_noise_scale=0.6, eos_threshold=None)


def make_problem(arch, B, P, G, seed=0, prompt_lens=None, **init_kw):
    cfg = oracle.get_arch(arch)
    init = dict(mod_gain=10.0, qk_gain=2.0)
    init.update(init_kw)
    sd = oracle.init_state_dict(cfg, seed=seed, **init)
    norm = {"center": sd["sonar_normalizer.center"], "scale": sd["sonar_normalizer.scale"]}
    x = synthetic_prompts(norm, B, P, seed=1234 + seed)
    if prompt_lens is not None:
        for i, n in enumerate(prompt_lens):
            x[i, n:] = 0
    steps = G + P
    nz = synthetic_noise(steps, B, seed=4321 + seed)
    return cfg, sd, x, nz


def run_oracle(cfg, sd, x, nz, G, opts, prompt_lens=None, eos_vec=None, trace=None):
    model = oracle.OracleTwoTower(cfg, sd)
    gen = oracle.OracleGenerator(model, oracle.GenOptions(**opts), eos_vec=eos_vec)
    lens = None if prompt_lens is None else torch.tensor(prompt_lens)
    return gen(x, prompt_lens=lens, max_gen_len=G, noise=nz, trace=trace)


def run_hip(arch, sd, x, nz, G, opts, dtype=torch.float32, prompt_lens=None, eos_vec=None, model=None):
    from lcm.datasets.batch import EmbeddingsBatch, PaddingMask
    from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs

    if model is None:
        model = create_two_tower_diffusion_lcm_model(_config(arch), device="cuda:0", dtype=dtype)
        model.load_state_dict(sd)
    gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**opts),
                                        eos_vec=None if eos_vec is None else eos_vec.cuda())
    mask = None
    if prompt_lens is not None:
        mask = PaddingMask(torch.tensor(prompt_lens, device="cuda:0"), x.shape[1])
    out = gen(EmbeddingsBatch(x.cuda().to(dtype), mask), max_gen_len=G, noise=nz.cuda())
    torch.cuda.current_stream().synchronize()     # not device-wide: other host threads may be recording a graph
    return out, model


def _config(arch):
    from lcm.models.two_tower_diffusion_lcm import lcm_archs
    if arch == "mini_two_tower_diffusion_lcm":
        return mini_config()
    return lcm_archs.get_config(arch)


def mini_config():
    """The `lcm`-side twin of oracle.archs._mini (a 1.6B-style model small enough for unit tests)."""
    from lcm.models.two_tower_diffusion_lcm import DenoiserConfig, TransformerConfig, TwoTowerDiffusionLCModelConfig
    from lcm.nn.schedulers import DDIMSchedulerConfig
    shared = dict(num_attn_heads=2, ffn_inner_dim=1024, use_swiglu=True, layer_normalization_style="rms",
                  mha_output_proj_bias=True)
    return TwoTowerDiffusionLCModelConfig(
        model_dim=256, max_seq_len=512,
        context_encoder=TransformerConfig(num_layers=2, pos_embedding_style="rope", **shared),
        denoiser=DenoiserConfig(num_layers=3, pos_embedding_style="none", timestep_embed_dim=256, **shared),
        noise_scheduler=DDIMSchedulerConfig(num_diffusion_train_steps=100),
        sonar_normalizer_name="dummy_sonar_normalizer", trained_with_cf_guidance=True)


def compare(hyps_hip, hyps_ref, sd, start=0):
    """Returns (max-abs in normalized space, min cosine per embedding, lengths equal) over all documents."""
    c, s = sd["sonar_normalizer.center"], sd["sonar_normalizer.scale"]
    worst, min_cos = 0.0, 1.0
    for hh, hr in zip(hyps_hip, hyps_ref):
        a = hh[0].seq.float().cpu()
        b = hr[0].seq
        assert a.shape == b.shape, (a.shape, b.shape)
        an, bn = (a - c) / s, (b - c) / s
        worst = max(worst, float((an[start:] - bn[start:]).abs().max()))
        cos = torch.nn.functional.cosine_similarity(an[start:], bn[start:], dim=-1)
        min_cos = min(min_cos, float(cos.min()))
    return worst, min_cos


# ---- base LCM ------------------------------------------------------------------------------------------------
def base_config(arch):
    """The `lcm`-side config of an oracle base arch (mini_base_lcm is the twin of oracle.archs._mini_base)."""
    from lcm.models.base_lcm import BaseLCModelConfig, TransformerConfig, lcm_archs
    if arch == "mini_base_lcm":
        return BaseLCModelConfig(
            model_dim=256, max_seq_len=512, sonar_normalizer_name="dummy_sonar_normalizer",
            lcm=TransformerConfig(num_layers=3, num_attn_heads=2, ffn_inner_dim=1024, use_swiglu=True,
                                  layer_normalization_style="rms", pos_embedding_style="rope", mha_output_proj_bias=True))
    return lcm_archs.get_config(arch)


def make_base_problem(arch, B, P, G, seed=0, prompt_lens=None, **init_kw):
    cfg = oracle.get_base_arch(arch)
    init = dict(qk_gain=2.0, pos_std=0.5)
    init.update(init_kw)
    sd = oracle.init_state_dict(cfg, seed=seed, **init)
    x = synthetic_prompts(oracle.load_dummy_normalizer(), B, P, seed=1234 + seed)
    if cfg.sonar_normalizer_name is None:
        x = x * 50.0      # no normalizer: bring the inputs to O(1) like normalized embeddings
    if prompt_lens is not None:
        for i, n in enumerate(prompt_lens):
            x[i, n:] = 0
    nz = synthetic_noise(G + P, B, seed=4321 + seed)
    return cfg, sd, x, nz


def run_oracle_base(cfg, sd, x, nz, G, opts, prompt_lens=None, eos_vec=None):
    gen = oracle.OracleGenerator(oracle.OracleBaseLCM(cfg, sd), oracle.GenOptions(**opts), eos_vec=eos_vec)
    lens = None if prompt_lens is None else torch.tensor(prompt_lens)
    return gen(x, prompt_lens=lens, max_gen_len=G, noise=nz)


def run_hip_base(arch, sd, x, nz, G, opts, dtype=torch.float32, prompt_lens=None, eos_vec=None, model=None,
                 disable_cache=False):
    from lcm.datasets.batch import EmbeddingsBatch, PaddingMask
    from lcm.inference.lcm import LCMGenerator, LCMGeneratorOptions
    from lcm.models.base_lcm import create_base_lcm_model

    if model is None:
        model = create_base_lcm_model(base_config(arch), device="cuda:0", dtype=dtype)
        model.load_state_dict(sd)
    gen = LCMGenerator(model, LCMGeneratorOptions(**opts), eos_vec=None if eos_vec is None else eos_vec.cuda())
    mask = None
    if prompt_lens is not None:
        mask = PaddingMask(torch.tensor(prompt_lens, device="cuda:0"), x.shape[1])
    out = gen(EmbeddingsBatch(x.cuda().to(dtype), mask), max_gen_len=G, noise=nz.cuda(), disable_cache=disable_cache)
    torch.cuda.current_stream().synchronize()     # not device-wide: other host threads may be recording a graph
    return out, model


def compare_raw(hyps_hip, hyps_ref, start=0):
    """(max-abs relative to the largest reference magnitude, min cosine) in the output space itself."""
    worst, min_cos = 0.0, 1.0
    for hh, hr in zip(hyps_hip, hyps_ref):
        a, b = hh[0].seq.float().cpu(), hr[0].seq
        assert a.shape == b.shape, (a.shape, b.shape)
        worst = max(worst, float((a[start:] - b[start:]).abs().max() / b[start:].abs().max()))
        min_cos = min(min_cos, float(torch.nn.functional.cosine_similarity(a[start:], b[start:], dim=-1).min()))
    return worst, min_cos
