"""Training-shaped forward of the two-tower diffusion LCM (all positions at once, per-position diffusion timesteps,
shifted causal cross-attention) against the CPU oracle's restatement of TwoTowerDiffusionLCModel.forward
(lcm/models/two_tower_diffusion_lcm/builder.py:128-159).  The reference holds no fixture for this path
("parity unpinned"); tolerances: fp32 max-abs <= 1e-3 and cosine >= 0.9999, bf16 cosine >= 0.999."""

import pytest
import torch

import oracle
from oracle.weights import synthetic_prompts

from parity_utils import _config

pytestmark = pytest.mark.gpu


def _problem(arch, B, T, seed, n_train, distinct=None, **init):
    cfg = oracle.get_arch(arch)
    sd = oracle.init_state_dict(cfg, seed=seed, **dict(dict(mod_gain=10.0, qk_gain=2.0), **init))
    norm = {"center": sd["sonar_normalizer.center"], "scale": sd["sonar_normalizer.scale"]}
    g = torch.Generator().manual_seed(99 + seed)
    x = (synthetic_prompts(norm, B, T, seed=7 + seed) - norm["center"]) / norm["scale"]       # normalized clean batch
    noisy = torch.randn(B, T, x.shape[-1], generator=g)
    if distinct is None:
        ts = torch.randint(0, n_train, (B, T), generator=g)
    else:
        pool = torch.randperm(n_train, generator=g)[:distinct]
        ts = pool[torch.randint(0, distinct, (B, T), generator=g)]
    return cfg, sd, x, noisy, ts


def _hip_forward(arch, sd, x, noisy, ts, dtype, lengths=None, source_lengths=None):
    from lcm.datasets.batch import EmbeddingsBatch, PaddingMask
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    model = create_two_tower_diffusion_lcm_model(_config(arch), device="cuda:0", dtype=dtype)
    model.load_state_dict(sd)
    mask = None if lengths is None else PaddingMask(torch.tensor(lengths, device="cuda:0"), x.shape[1])
    src = None if source_lengths is None else torch.tensor(source_lengths, device="cuda:0")
    batch = EmbeddingsBatch(x.cuda().to(dtype), mask, source_lengths=src)
    noisy_batch = EmbeddingsBatch(noisy.cuda(), mask, diffusion_timesteps=ts.cuda())
    out = model.forward(batch, noisy_batch)
    torch.cuda.synchronize()
    return out.seqs.float().cpu(), model


def _check(got, want, lengths, max_abs, min_cos):
    B, T, _ = want.shape
    for i in range(B):
        n = T if lengths is None else lengths[i]
        a, b = got[i, :n], want[i, :n]
        assert torch.isfinite(got[i]).all()
        cos = torch.nn.functional.cosine_similarity(a, b, dim=-1).min().item()
        err = (a - b).abs().max().item()
        assert cos >= min_cos and (max_abs is None or err <= max_abs), (i, cos, err)


@pytest.mark.parametrize("arch,n_train", [("toy_two_tower_diffusion_lcm", 1000), ("mini_two_tower_diffusion_lcm", 100)])
def test_forward_matches_oracle_fp32(arch, n_train):
    cfg, sd, x, noisy, ts = _problem(arch, 3, 7, 0, n_train)
    want = oracle.OracleTwoTower(cfg, sd).forward(x, noisy, ts)
    got, _ = _hip_forward(arch, sd, x, noisy, ts, torch.float32)
    _check(got, want, None, 1e-3, 0.9999)


def test_forward_padding_and_source_lengths_fp32():
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, noisy, ts = _problem(arch, 4, 9, 1, 100)
    lengths, src = [9, 5, 7, 2], [0, 2, 3, 1]
    want = oracle.OracleTwoTower(cfg, sd).forward(x, noisy, ts, lengths=torch.tensor(lengths), source_lengths=torch.tensor(src))
    got, _ = _hip_forward(arch, sd, x, noisy, ts, torch.float32, lengths=lengths, source_lengths=src)
    _check(got, want, lengths, 1e-3, 0.9999)


def test_forward_many_rows_bf16():
    """Enough rows for the matrix-core GEMM and the matrix-core causal attention kernel, several passes
    (document groups) inside one call; bf16 against the fp32 oracle on bf16-rounded weights."""
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, noisy, ts = _problem(arch, 40, 130, 2, 100, distinct=20, mod_gain=4.0, qk_gain=1.0)
    sd = {k: v.bfloat16().float() for k, v in sd.items()}
    lengths = [130 - (3 * i) % 60 for i in range(40)]
    want = oracle.OracleTwoTower(cfg, sd).forward(x, noisy, ts, lengths=torch.tensor(lengths))
    got, _ = _hip_forward(arch, sd, x, noisy, ts, torch.bfloat16, lengths=lengths)
    _check(got, want, lengths, None, 0.999)


def test_forward_last_position_equals_one_sampling_step_input():
    """With one timestep everywhere, position j of the all-positions pass is the denoiser call the sampling loop
    makes after j context positions: compare position T-1 with a direct oracle denoise on the prefix."""
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, noisy, _ = _problem(arch, 2, 6, 3, 100)
    ts = torch.full((2, 6), 57)
    got, _ = _hip_forward(arch, sd, x, noisy, ts, torch.float32)
    m = oracle.OracleTwoTower(cfg, sd)
    ctx = m.encode(x[:, :5])
    want = m.denoise(noisy[:, 5:6], ts[:, 5:6], ctx)
    assert (got[:, 5:6] - want).abs().max().item() <= 1e-3


@pytest.mark.timeout(900)
def test_forward_1_6b_bf16():
    """The 1.6B architecture in bf16 with 8 x 160 = 1280 rows: every projection of the pass takes the matrix-core
    many-row GEMM, the cross-attention the matrix-core causal kernel.  Milder gains, as for the 1.6B sampling test."""
    arch = "two_tower_diffusion_lcm_1_6B"
    cfg, sd, x, noisy, ts = _problem(arch, 8, 160, 4, 100, distinct=30, mod_gain=4.0, qk_gain=1.0)
    sd = {k: v.bfloat16().float() for k, v in sd.items()}
    lengths = [160, 160, 97, 130, 160, 33, 160, 150]
    want = oracle.OracleTwoTower(cfg, sd).forward(x.bfloat16().float(), noisy, ts, lengths=torch.tensor(lengths))
    got, model = _hip_forward(arch, sd, x, noisy, ts, torch.bfloat16, lengths=lengths)
    _check(got, want, lengths, None, 0.999)
    del model


def test_model_denoise_and_guidance_methods():
    """`TwoTowerDiffusionLCModel.denoise` (builder.py:193-213) on the context handle of `encode`: with `inference` every
    latent sees the whole context (the sampling loop's call, here with three latents per document at different
    timesteps), without it the shifted causal mask; both against the oracle's denoiser, and the sampling loop still
    runs on the same session afterwards.  `apply_classifier_free_guidance` (builder.py:463-488) against the oracle."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    from lcm.nn.incremental_state import LCMIncrementalStateBag
    from parity_utils import DEFAULT_OPTS, make_problem, mini_config
    O = oracle
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 3, 6, 1, mod_gain=4.0)
    om = O.OracleTwoTower(cfg, sd)
    opts = O.GenOptions(**dict(DEFAULT_OPTS, inference_timesteps=10))
    om.prep_for_denoising(opts)
    model = create_two_tower_diffusion_lcm_model(mini_config(), device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    model.prep_for_denoising(DiffusionLCMGeneratorOptions(**dict(DEFAULT_OPTS, inference_timesteps=10)))
    xn = om.normalize(x)
    g = torch.Generator().manual_seed(5)
    noisy = torch.randn(3, 3, 1024, generator=g)
    ts = torch.tensor([[90, 17, 3], [55, 55, 0], [99, 42, 42]])
    ctx_ref = om.encode(xn)
    bag = LCMIncrementalStateBag(16)
    ctx = model.encode(EmbeddingsBatch(xn.cuda(), None), state_bag=bag)
    nb = EmbeddingsBatch(noisy.cuda(), None, diffusion_timesteps=ts.cuda())
    # inference: no cross-attention mask
    ref = om.denoise(noisy, ts, ctx_ref)
    out = model.denoise(nb, ctx, inference=True).seqs.cpu()
    assert (out - ref).abs().max() <= 1e-3 * max(1.0, float(ref.abs().max())), float((out - ref).abs().max())
    # training shape: position j sees [zero vector, context[: j + source_length]]
    src = torch.tensor([2, 0, 3])
    mask = torch.stack([torch.ones(3, 7).tril_(diagonal=int(n)) for n in src]).unsqueeze(1).log()
    ref2 = om.denoise(noisy, ts, ctx_ref, cross_mask=mask)
    out2 = model.denoise(nb, ctx, source_lengths=src.cuda()).seqs.cpu()
    assert (out2 - ref2).abs().max() <= 1e-3 * max(1.0, float(ref2.abs().max()))
    assert (out2 - out).abs().max() > 1e-3          # the masks differ
    # the sampling loop on the same session afterwards uses its own schedule again
    nxt, _ = model.predict_next_sentence(EmbeddingsBatch(x[:, -1:].cuda(), None), ctx, state_bag=LCMIncrementalStateBag(16),
                                         context_state_bag=bag, noise=nz[0].cuda())
    assert torch.isfinite(nxt.seqs).all()
    with pytest.raises(NotImplementedError):
        model.denoise(nb, ctx, cf_guidance_prob=0.1)
    # guidance
    eps = torch.randn(8, 2, 1024, generator=g)
    want = om.apply_cfg(eps)
    got = model.apply_classifier_free_guidance(eps.cuda()).cpu()
    assert got.shape == want.shape and (got - want).abs().max() <= 2e-5 * max(1.0, float(want.abs().max()))
    del model
