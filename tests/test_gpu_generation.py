"""End-to-end parity of the HIP generation path against the CPU oracle on identical seeded
weights, prompts and host-supplied noise.

Tolerances: fp32 -- max-abs <= 1e-3 (in normalizer units, i.e. the space
the model works in) and per-embedding cosine >= 0.9999; bf16 -- cosine >= 0.999.
"""

import pytest
import torch

from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions
import oracle as oracle_mod
from parity_utils import DEFAULT_OPTS, compare, make_problem, run_hip, run_oracle

pytestmark = pytest.mark.gpu

FP32_MAX_ABS, FP32_COS, BF16_COS = 1e-3, 0.9999, 0.999


def test_toy_fp32_config1():
    """The CPU-runnable benchmark configuration: toy arch, 8 docs x (4 prompt + 12 generated), 10 DDIM steps, CFG."""
    arch = "toy_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 8, 4, 12)
    opts = dict(DEFAULT_OPTS, inference_timesteps=10)
    ref = run_oracle(cfg, sd, x, nz, 12, opts)
    out, _ = run_hip(arch, sd, x, nz, 12, opts)
    worst, cos = compare(out.hypotheses, ref, sd, start=4)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)
    assert all(h[0].seq.shape == (16, 1024) for h in out.hypotheses)


@pytest.mark.parametrize("guidance", [1.0, 3.0])
def test_mini_fp32(guidance):
    """1.6B-style blocks (RMSNorm, RoPE, SwiGLU, out-proj bias, T=100 -> 50 executed steps)."""
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 5, 3, 4)
    opts = dict(DEFAULT_OPTS, inference_timesteps=40, guidance_scale=guidance)
    ref = run_oracle(cfg, sd, x, nz, 4, opts)
    out, _ = run_hip(arch, sd, x, nz, 4, opts)
    worst, cos = compare(out.hypotheses, ref, sd, start=3)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)


def test_mini_bf16():
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 6, 3, 4)
    opts = dict(DEFAULT_OPTS, inference_timesteps=40)
    # the oracle sees the bf16-rounded weights and prompts, in fp32 arithmetic
    sd16 = {k: v.bfloat16().float() for k, v in sd.items()}
    ref = run_oracle(cfg, sd16, x.bfloat16().float(), nz, 4, opts)
    out, _ = run_hip(arch, sd, x, nz, 4, opts, dtype=torch.bfloat16)
    worst, cos = compare(out.hypotheses, ref, sd16, start=3)
    assert cos >= BF16_COS, (worst, cos)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mini_long_ragged_prompts(dtype):
    """Long contexts (the long-context configuration's regime at a size the oracle finishes quickly): prompts of
    300 / 283 / 300 sentences, so the prefill runs several 64-query blocks per document, decoding attends over
    ~300 cached positions, and the cross-attention over 301 keys (many rounds of 16 plus a ragged tail)."""
    arch = "mini_two_tower_diffusion_lcm"
    lens = [300, 283, 300]
    cfg, sd, x, nz = make_problem(arch, 3, 300, 2, prompt_lens=lens, mod_gain=4.0, qk_gain=1.0)
    opts = dict(DEFAULT_OPTS, inference_timesteps=40)
    if dtype == torch.bfloat16:
        sd = {k: v.bfloat16().float() for k, v in sd.items()}
        x = x.bfloat16().float()
    ref = run_oracle(cfg, sd, x, nz, 2, opts, prompt_lens=lens)
    out, _ = run_hip(arch, sd, x, nz, 2, opts, dtype=dtype, prompt_lens=lens)
    assert [h[0].seq.shape[0] for h in out.hypotheses] == [n + 2 for n in lens]
    worst, cos = 0.0, 1.0
    for n, hh, hr in zip(lens, out.hypotheses, ref):
        w, c = compare([hh], [hr], sd, start=n)
        worst, cos = max(worst, w), min(cos, c)
    if dtype == torch.float32:
        assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)
    else:
        assert cos >= BF16_COS, (worst, cos)


def test_mini_generation_at_the_context_limit():
    """A session's longest context: 3800 positions.  Prompts of 3792 / 3700 sentences + 6 generated (prefill in several
    passes of the many-row kernels, decode attention and cross-attention over ~3800 keys) against the oracle, fp32; one
    position more is refused by name."""
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    from oracle.weights import synthetic_noise, synthetic_prompts
    from parity_utils import mini_config
    lens, G = [3792, 3700], 6
    cfg = oracle_mod.get_arch("mini_two_tower_diffusion_lcm")
    cfg.max_seq_len = 4096
    sd = oracle_mod.init_state_dict(cfg, seed=3, mod_gain=4.0, qk_gain=1.0)
    x = synthetic_prompts({"center": sd["sonar_normalizer.center"], "scale": sd["sonar_normalizer.scale"]}, 2, max(lens), seed=1237)
    x[1, lens[1]:] = 0
    nz = synthetic_noise(G + max(lens), 2, seed=4324)
    opts = dict(DEFAULT_OPTS, inference_timesteps=10)
    ref = run_oracle(cfg, sd, x, nz, G, opts, prompt_lens=lens)
    mc = mini_config()
    mc.max_seq_len = 4096
    model = create_two_tower_diffusion_lcm_model(mc, device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    out, _ = run_hip("mini_two_tower_diffusion_lcm", sd, x, nz, G, opts, prompt_lens=lens, model=model)
    assert [h[0].seq.shape[0] for h in out.hypotheses] == [n + G for n in lens]
    for n, hh, hr in zip(lens, out.hypotheses, ref):
        worst, cos = compare([hh], [hr], sd, start=n)
        assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (n, worst, cos)
    with pytest.raises(RuntimeError, match="3800"):
        run_hip("mini_two_tower_diffusion_lcm", sd, x, nz, G + 3, opts, prompt_lens=lens, model=model)
    del model


def test_toy_every_training_step_as_a_sampling_step():
    """`inference_timesteps` = the scheduler's 1000 training steps: the longest schedule the toy arch can ask for (a
    modulation table of 1000 x 2 layers, a recorded loop of ~22 000 kernel nodes), one sentence, against the oracle."""
    arch = "toy_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 2, 3, 1)
    opts = dict(DEFAULT_OPTS, inference_timesteps=1000)
    ref = run_oracle(cfg, sd, x, nz, 1, opts)
    out, model = run_hip(arch, sd, x, nz, 1, opts)
    assert len(model.noise_scheduler.timesteps) == 1000
    worst, cos = compare(out.hypotheses, ref, sd, start=3)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)


def test_ragged_prompts_and_early_finish():
    """Ragged prompts: rows still inside their prompt keep the prompt embedding, documents finish at
    different steps (prompt_len + max_gen_len) and leave the batch."""
    arch = "toy_two_tower_diffusion_lcm"
    lens = [2, 5, 3, 5]
    cfg, sd, x, nz = make_problem(arch, 4, 5, 3, prompt_lens=lens)
    opts = dict(DEFAULT_OPTS, inference_timesteps=5)
    ref = run_oracle(cfg, sd, x, nz, 3, opts, prompt_lens=lens)
    out, _ = run_hip(arch, sd, x, nz, 3, opts, prompt_lens=lens)
    for i, n in enumerate(lens):
        assert out.hypotheses[i][0].seq.shape[0] == n + 3 == ref[i][0].seq.shape[0]
    worst, cos = compare(out.hypotheses, ref, sd)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)


def test_eos_stop_matches_oracle():
    """An EOS vector equal to document 1's second generated sentence stops that document there."""
    arch = "toy_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 3, 3, 5)
    opts = dict(DEFAULT_OPTS, inference_timesteps=5)
    free = run_oracle(cfg, sd, x, nz, 5, opts)
    eos = free[1][0].seq[4].clone()
    opts_eos = dict(opts, eos_threshold=0.999, trim_hypotheses=True)
    ref = run_oracle(cfg, sd, x, nz, 5, opts_eos, eos_vec=eos)
    out, _ = run_hip(arch, sd, x, nz, 5, opts_eos, eos_vec=eos)
    assert [h[0].seq.shape[0] for h in out.hypotheses] == [h[0].seq.shape[0] for h in ref]
    assert ref[1][0].seq.shape[0] == 4          # stopped at step 4, EOS trimmed
    worst, cos = compare(out.hypotheses, ref, sd)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)


def test_batch_of_one_equals_batch(monkeypatch):
    """The reference's batched-vs-single invariant (tests/units/inference/
    test_base_lcm_batched_inference.py:21-96) on the HIP path: a document generated alone equals
    the same document generated inside a batch (noise indexed by document)."""
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 3, 3, 3)
    opts = dict(DEFAULT_OPTS, inference_timesteps=10)
    out, model = run_hip(arch, sd, x, nz, 3, opts)
    for i in range(3):
        single, _ = run_hip(arch, sd, x[i:i + 1], nz[:, i:i + 1], 3, opts, model=model)
        a, b = single.hypotheses[0][0].seq, out.hypotheses[i][0].seq
        assert (a - b).abs().max() <= 1e-6 * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_document_shards_equal_one_batch(dtype):
    """What `bench.py --gpus N` relies on, in one process: 8 documents generated as one batch, as 2 shards of 4 and
    as 4 shards of 2 (cut by `shard_bounds`, prompts and noise taken by GLOBAL document id, concatenated in rank
    order like `all_gather_generations` does) give the same generations -- fp32 to 1e-3 / cosine 0.9999, bf16 to
    the bf16 bar (shards of another size take other tile plans).  Ragged prompts and an EOS vector, so documents
    also finish at different steps inside every shard.  Only the RCCL call itself is left to a multi-GPU run."""
    from lcm.utils.distributed import shard_bounds
    arch = "mini_two_tower_diffusion_lcm"
    lens = [3, 4, 2, 4, 4, 3, 4, 2]
    cfg, sd, x, nz = make_problem(arch, 8, 4, 4, prompt_lens=lens)
    if dtype == torch.bfloat16:
        sd = {k: oracle_mod.r16(v) for k, v in sd.items()}
        x = oracle_mod.r16(x)
    opts = dict(DEFAULT_OPTS, inference_timesteps=10, eos_threshold=0.6)
    ref = run_oracle(cfg, sd, x, nz, 4, dict(opts, eos_threshold=None), prompt_lens=lens)
    eos = ref[5][0].seq[5]                       # a sentence document 5 will generate: it stops early there
    whole, model = run_hip(arch, sd, x, nz, 4, opts, dtype=dtype, prompt_lens=lens, eos_vec=eos)
    n_finished_early = sum(h[0].seq.shape[0] < n + 4 for h, n in zip(whole.hypotheses, lens))
    assert n_finished_early >= 1
    for world in (2, 4):
        got = []
        for rank in range(world):
            lo, hi = shard_bounds(8, rank, world)
            # the generator consumes noise[step - shortest prompt OF ITS BATCH, document]: a shard whose shortest prompt
            # is longer starts later in the global table (draws belong to (position, global document id))
            first = min(lens[lo:hi]) - min(lens)
            part, _ = run_hip(arch, sd, x[lo:hi], nz[first:, lo:hi], 4, opts, dtype=dtype, prompt_lens=lens[lo:hi], eos_vec=eos,
                              model=model)
            got += part.hypotheses
        assert len(got) == 8
        for i, (hs, hw) in enumerate(zip(got, whole.hypotheses)):
            a, b = hs[0].seq.float(), hw[0].seq.float()
            assert a.shape == b.shape, (world, i, a.shape, b.shape)
            an, bn = [(t.cpu() - sd["sonar_normalizer.center"]) / sd["sonar_normalizer.scale"] for t in (a, b)]
            cos = float(torch.nn.functional.cosine_similarity(an, bn, dim=-1).min())
            if dtype == torch.float32:
                assert float((an - bn).abs().max()) <= FP32_MAX_ABS and cos >= FP32_COS, (world, i, cos)
            else:
                assert cos >= BF16_COS, (world, i, cos)
    del model


def test_session_pool_holds_one_workspace_at_a_time():
    """A call that needs a larger session than the pooled one drops the pooled workspace BEFORE it allocates the new
    one (at 2048 documents x 500 positions a workspace is 142 GB: two of them do not fit)."""
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 6, 40, 8)
    opts = dict(DEFAULT_OPTS, inference_timesteps=10)
    _, model = run_hip(arch, sd, x, nz, 1, opts)
    w1 = model._session_pool[0].workspace.numel()
    torch.cuda.synchronize()
    a0 = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    run_hip(arch, sd, x, nz, 8, opts, model=model)            # more positions: the pooled session is too small
    w2 = model._session_pool[0].workspace.numel()
    assert w2 > w1 > (1 << 20)
    peak = torch.cuda.max_memory_allocated()
    assert peak < a0 + w2 - w1 // 2, (a0, w1, w2, peak)        # both alive at once would be >= a0 + w2
    del model


def test_no_cpu_fallback():
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        create_two_tower_diffusion_lcm_model(lcm_archs.get_config("toy_two_tower_diffusion_lcm"), device="cpu")


def test_toy_fp32_against_golden_fixture():
    """The committed fixture (tests/golden/toy_two_tower_golden.pt, written by oracle/make_golden.py) holds
    prompts, noise and the oracle's outputs: this test needs no oracle arithmetic at run time beyond the
    seeded weight initialisation."""
    import os

    import oracle
    blob = torch.load(os.path.join(os.path.dirname(__file__), "golden", "toy_two_tower_golden.pt"), weights_only=True)
    sd = oracle.init_state_dict(oracle.get_arch(blob["arch"]), **blob["init"])
    out, model = run_hip(blob["arch"], sd, blob["prompts"], blob["noise"], blob["G"], blob["options"])
    got = torch.stack([h[0].seq.float().cpu() for h in out.hypotheses])
    c, s = sd["sonar_normalizer.center"], sd["sonar_normalizer.scale"]
    diff = ((got - blob["outputs"]) / s).abs().max()
    assert diff <= FP32_MAX_ABS, float(diff)
    # per-step latents of the first generated sentence through predict_next_sentence(trace=...)
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.nn.incremental_state import LCMIncrementalStateBag
    P, B, opts = blob["P"], blob["B"], blob["options"]
    model.prep_for_denoising(DiffusionLCMGeneratorOptions(**opts))
    bag = LCMIncrementalStateBag(P + 1)
    x = blob["prompts"].cuda()
    ctx = model.encode(EmbeddingsBatch(x[:, :P - 1]), bag, normalized=False)
    bag.increment_step_nr(P - 1)
    steps = len(model.noise_scheduler.timesteps)
    trace = torch.zeros(steps, B, 1024, device="cuda:0")
    model.predict_next_sentence(EmbeddingsBatch(x[:, P - 1:P]), ctx, context_state_bag=bag, noise=blob["noise"][0].cuda(),
                                trace=trace)
    lat = trace.cpu()
    for idx, key in [(0, "latents_step0"), (1, "latents_step1"), (-1, "latents_last")]:
        want = blob[key].reshape(lat[idx].shape)
        assert (lat[idx] - want).abs().max() <= FP32_MAX_ABS, key


@pytest.mark.timeout(900)
def test_1_6b_bf16_reduced_batch():
    """The headline configuration's architecture (1.6B, bf16, 40 requested -> 50 executed steps, CFG 3.0 / 0.7) at
    a batch the fp32 CPU oracle finishes in about a minute: 4 docs x (3 prompt + 2 generated).

    The random weights use milder gains than the small-model tests (modulator x4, q/k x1): with the x10 / x2
    gains a 13-layer random denoiser amplifies bf16 rounding of the activations to cosine 0.9990 after one
    sentence and 0.985 after two (fp32 on the same weights stays within 6e-5 of the oracle), which measures
    the conditioning of that random network rather than the kernels."""
    arch = "two_tower_diffusion_lcm_1_6B"
    cfg, sd, x, nz = make_problem(arch, 4, 3, 2, mod_gain=4.0, qk_gain=1.0)
    opts = dict(DEFAULT_OPTS, inference_timesteps=40)
    sd = {k: v.bfloat16().float() for k, v in sd.items()}
    ref = run_oracle(cfg, sd, x.bfloat16().float(), nz, 2, opts)
    out, model = run_hip(arch, sd, x, nz, 2, opts, dtype=torch.bfloat16)
    worst, cos = compare(out.hypotheses, ref, sd, start=3)
    assert cos >= BF16_COS, (worst, cos)
    del model, out


@pytest.mark.timeout(900)
def test_1_6b_fp32_reduced_batch():
    """Same architecture in fp32 with the strong gains: the kernels themselves track the oracle to the
    fp32 bar through 13 layers x 50 steps x 2 sentences."""
    arch = "two_tower_diffusion_lcm_1_6B"
    cfg, sd, x, nz = make_problem(arch, 3, 3, 2)
    opts = dict(DEFAULT_OPTS, inference_timesteps=40)
    ref = run_oracle(cfg, sd, x, nz, 2, opts)
    out, model = run_hip(arch, sd, x, nz, 2, opts)
    worst, cos = compare(out.hypotheses, ref, sd, start=3)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)
    del model, out


@pytest.mark.timeout(900)
def test_1_6b_full_batch_properties():
    """At the benchmark's batch (64 docs, bf16, 1.6B) the oracle is too slow, so size-independent
    properties stand in: (1) the run is deterministic bit for bit; (2) a document generated alone matches
    the same document inside the batch of 64 (different GEMM tile plans, so bf16 tolerance); (3) guidance
    scale 1 equals the unguided conditional path (the reference skips CFG arithmetic only at w == 1, where
    cond + 1 * (cond - uncond) - ... reduces to cond)."""
    import oracle
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs
    arch = "two_tower_diffusion_lcm_1_6B"
    cfg = oracle.get_arch(arch)
    model = create_two_tower_diffusion_lcm_model(lcm_archs.get_config(arch), device="cuda:0", dtype=torch.bfloat16)
    model.init_parameters(seed=3, modulator_std=0.02)
    norm = oracle.load_dummy_normalizer()
    from oracle.weights import synthetic_noise, synthetic_prompts
    B, P, G = 64, 8, 3
    x = synthetic_prompts(norm, B, P, seed=11)
    nz = synthetic_noise(P + G, B, seed=12)
    opts = dict(DEFAULT_OPTS, inference_timesteps=40)
    sd = None
    a, _ = run_hip(arch, sd, x, nz, G, opts, dtype=torch.bfloat16, model=model)
    b, _ = run_hip(arch, sd, x, nz, G, opts, dtype=torch.bfloat16, model=model)
    for ha, hb in zip(a.hypotheses, b.hypotheses):
        assert ha[0].seq.shape == (P + G, 1024) and torch.equal(ha[0].seq, hb[0].seq)
        assert torch.isfinite(ha[0].seq.float()).all()
    for i in (0, 37, 63):
        single, _ = run_hip(arch, sd, x[i:i + 1], nz[:, i:i + 1], G, opts, dtype=torch.bfloat16, model=model)
        u, v = single.hypotheses[0][0].seq.float()[P:], a.hypotheses[i][0].seq.float()[P:]
        assert torch.nn.functional.cosine_similarity(u, v, dim=-1).min() >= BF16_COS


@pytest.mark.timeout(900)
@pytest.mark.parametrize("B", [200, 512])
def test_1_6b_many_documents_properties(B):
    """Many documents per GPU (the long-context configuration's batch regime): with 400 / 1024 denoiser rows the
    products run on the matrix-core kernels (128 x 128 with split-K slabs, 256 x 256 for the SwiGLU projection) or,
    for the conditional rows only, on several row tiles of the weight-streaming kernel.  Same size-independent
    properties as at 64 documents: bit-for-bit determinism, and a document generated alone (64-row weight-streaming
    tile) matches the same document inside the batch to the bf16 bar."""
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs
    from oracle.weights import synthetic_noise, synthetic_prompts
    arch = "two_tower_diffusion_lcm_1_6B"
    model = create_two_tower_diffusion_lcm_model(lcm_archs.get_config(arch), device="cuda:0", dtype=torch.bfloat16)
    model.init_parameters(seed=3, modulator_std=0.02)
    P, G = 4, 2
    x = synthetic_prompts(oracle_mod.load_dummy_normalizer(), B, P, seed=21)
    nz = synthetic_noise(P + G, B, seed=22)
    opts = dict(DEFAULT_OPTS, inference_timesteps=40)
    a, _ = run_hip(arch, None, x, nz, G, opts, dtype=torch.bfloat16, model=model)
    b, _ = run_hip(arch, None, x, nz, G, opts, dtype=torch.bfloat16, model=model)
    for ha, hb in zip(a.hypotheses, b.hypotheses):
        assert ha[0].seq.shape == (P + G, 1024) and torch.equal(ha[0].seq, hb[0].seq)
        assert torch.isfinite(ha[0].seq.float()).all()
    for i in (0, B // 2 + 1, B - 1):
        single, _ = run_hip(arch, None, x[i:i + 1], nz[:, i:i + 1], G, opts, dtype=torch.bfloat16, model=model)
        u, v = single.hypotheses[0][0].seq.float()[P:], a.hypotheses[i][0].seq.float()[P:]
        assert torch.nn.functional.cosine_similarity(u, v, dim=-1).min() >= BF16_COS
    del model


def test_graph_replay_equals_eager():
    """On a non-default stream the denoising loop is recorded into a graph once per active-document count and
    replayed; the outputs must equal the eager launches bit for bit, also when documents leave the batch
    (ragged prompts -> different finishing steps -> several recorded loops)."""
    arch = "mini_two_tower_diffusion_lcm"
    lens = [2, 4, 3, 4]
    cfg, sd, x, nz = make_problem(arch, 4, 4, 3, prompt_lens=lens)
    opts = dict(DEFAULT_OPTS, inference_timesteps=10)
    from lcm import _hip
    from parity_utils import _config
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    stream = torch.cuda.Stream()
    model = create_two_tower_diffusion_lcm_model(_config(arch), device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    model.session_flags = _hip.SESSION_NO_GRAPH        # sessions that launch the loop kernel by kernel
    with torch.cuda.stream(stream):
        eager, _ = run_hip(arch, sd, x, nz, 3, opts, prompt_lens=lens, model=model)
    model.session_flags = 0
    with torch.cuda.stream(stream):
        graph, _ = run_hip(arch, sd, x, nz, 3, opts, prompt_lens=lens, model=model)
        again, _ = run_hip(arch, sd, x, nz, 3, opts, prompt_lens=lens, model=model)     # replays of cached graphs
    for he, hg, ha in zip(eager.hypotheses, graph.hypotheses, again.hypotheses):
        assert torch.equal(he[0].seq, hg[0].seq) and torch.equal(he[0].seq, ha[0].seq)
    ref = run_oracle(cfg, sd, x, nz, 3, opts, prompt_lens=lens)
    worst, cos = compare(graph.hypotheses, ref, sd)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)


@pytest.mark.timeout(900)
def test_7b_arch_runs_and_matches_batch_of_one():
    """two_tower_diffusion_lcm_7B (d = 4096, 32 heads, 5 + 14 layers, FFN 10 944 -> 11 008): too large for the CPU
    oracle inside a test, so the size-independent properties stand in: finite outputs, bit-exact repeatability,
    and a document alone == the same document in a batch."""
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs
    import oracle
    from oracle.weights import synthetic_noise, synthetic_prompts
    arch = "two_tower_diffusion_lcm_7B"
    model = create_two_tower_diffusion_lcm_model(lcm_archs.get_config(arch), device="cuda:0", dtype=torch.bfloat16)
    model.init_parameters(seed=5, modulator_std=0.02)
    assert model.num_parameters() == 6_930_781_696        # the count stated in the reference's archs.py docstring
    B, P, G = 6, 3, 2
    x = synthetic_prompts(oracle.load_dummy_normalizer(), B, P, seed=21)
    nz = synthetic_noise(P + G, B, seed=22)
    opts = dict(DEFAULT_OPTS, inference_timesteps=40)
    a, _ = run_hip(arch, None, x, nz, G, opts, dtype=torch.bfloat16, model=model)
    b, _ = run_hip(arch, None, x, nz, G, opts, dtype=torch.bfloat16, model=model)
    for ha, hb in zip(a.hypotheses, b.hypotheses):
        assert ha[0].seq.shape == (P + G, 1024) and torch.equal(ha[0].seq, hb[0].seq)
        assert torch.isfinite(ha[0].seq.float()).all()
    one, _ = run_hip(arch, None, x[2:3], nz[:, 2:3], G, opts, dtype=torch.bfloat16, model=model)
    u, v = one.hypotheses[0][0].seq.float()[P:], a.hypotheses[2][0].seq.float()[P:]
    assert torch.nn.functional.cosine_similarity(u, v, dim=-1).min() >= BF16_COS


def test_single_sentence_prompt_and_length_asserts():
    """A one-sentence prompt has nothing to prefill (reference generator.py:244-259 logs and skips); the length
    assertions of the reference (lcm/inference/lcm/generator.py:161-172) are kept."""
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 3, 1, 3)
    opts = dict(DEFAULT_OPTS, inference_timesteps=10)
    ref = run_oracle(cfg, sd, x, nz, 3, opts)
    out, model = run_hip(arch, sd, x, nz, 3, opts)
    worst, cos = compare(out.hypotheses, ref, sd, start=1)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)
    with pytest.raises(AssertionError, match="Generator can generate up to"):
        run_hip(arch, sd, x, nz, 1000, opts, model=model)
    from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator
    with pytest.raises(AssertionError, match="min_seq_len"):
        TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(min_seq_len=0))


def test_min_gen_len_repetition_stop_and_eos_token_kept():
    """Stopping rules of finalize_step (reference generator.py:331-372, 425-440): an EOS hit before min_gen_len is
    ignored; a repetition threshold of -1 stops every document at its first allowed step; include_eos_token keeps
    the stopping embedding in a trimmed hypothesis."""
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 2, 2, 5)
    base = dict(DEFAULT_OPTS, inference_timesteps=5)
    free = run_oracle(cfg, sd, x, nz, 5, base)
    eos = free[0][0].seq[2].clone()                       # document 0's first generated sentence
    for extra, min_len in [(dict(eos_threshold=0.999, trim_hypotheses=True), 1),
                           (dict(eos_threshold=0.999, trim_hypotheses=True), 3),
                           (dict(eos_threshold=0.999, trim_hypotheses=True, include_eos_token=True), 1),
                           (dict(stop_on_repetition_cosine_threshold=-1.0, min_seq_len=2), None)]:
        o = dict(base, **extra)
        kw = {} if min_len is None else dict(min_gen_len=min_len)
        model_o = oracle_mod.OracleTwoTower(cfg, sd)
        gen_o = oracle_mod.OracleGenerator(model_o, oracle_mod.GenOptions(**o), eos_vec=eos)
        ref = gen_o(x, max_gen_len=5, noise=nz, **kw)
        from lcm.datasets.batch import EmbeddingsBatch
        from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator
        from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
        from parity_utils import mini_config
        model = create_two_tower_diffusion_lcm_model(mini_config(), device="cuda:0", dtype=torch.float32)
        model.load_state_dict(sd)
        gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**o), eos_vec=eos.cuda())
        out = gen(EmbeddingsBatch(x.cuda(), None), max_gen_len=5, noise=nz.cuda(), **kw)
        assert [h[0].seq.shape[0] for h in out.hypotheses] == [h[0].seq.shape[0] for h in ref], (extra, min_len)
        worst, cos = compare(out.hypotheses, ref, sd)
        assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (extra, worst, cos)


def test_session_capacity_is_checked():
    """Appending beyond the session's context capacity is an error, not an out-of-bounds write."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    from lcm.nn.incremental_state import LCMIncrementalStateBag
    from parity_utils import mini_config
    model = create_two_tower_diffusion_lcm_model(mini_config(), device="cuda:0", dtype=torch.float32)
    model.prep_for_denoising(DiffusionLCMGeneratorOptions(inference_timesteps=5))
    bag = LCMIncrementalStateBag(4)
    model.encode(EmbeddingsBatch(torch.randn(2, 3, 1024, device="cuda"), None), bag)
    with pytest.raises(RuntimeError, match="max_ctx"):
        model.encode(EmbeddingsBatch(torch.randn(2, 2, 1024, device="cuda"), None), bag)
    model.release_session(bag.session)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_stopping_patterns(seed):
    """Ragged prompts plus an EOS vector that random documents hit at random steps: rows leave the batch one by one
    (slot-map updates, a recorded loop per active count on a non-default stream); lengths and embeddings must follow
    the oracle throughout."""
    arch = "mini_two_tower_diffusion_lcm"
    g = torch.Generator().manual_seed(100 + seed)
    B, P, G = 7, 4, 6
    lens = torch.randint(1, P + 1, (B,), generator=g).tolist()
    lens[int(torch.randint(0, B, (1,), generator=g))] = P
    cfg, sd, x, nz = make_problem(arch, B, P, G, seed=seed, prompt_lens=lens)
    opts = dict(DEFAULT_OPTS, inference_timesteps=5)
    free = run_oracle(cfg, sd, x, nz, G, opts, prompt_lens=lens)
    # an EOS direction close to several generated sentences: their mean
    picks = [free[i][0].seq[lens[i] + int(torch.randint(0, G, (1,), generator=g))] for i in range(B)]
    eos = torch.stack(picks).mean(0)
    sims = torch.cat([torch.nn.functional.cosine_similarity(eos[None], free[i][0].seq[lens[i]:]) for i in range(B)])
    srt = sims.sort().values
    k = int(0.6 * len(srt))
    thr = float((srt[k] + srt[k + 1]) / 2)     # ~40 % of the free-running sentences would stop; no value sits on the threshold
    o = dict(opts, eos_threshold=thr, trim_hypotheses=bool(seed % 2))
    ref = run_oracle(cfg, sd, x, nz, G, o, prompt_lens=lens, eos_vec=eos)
    with torch.cuda.stream(torch.cuda.Stream()):
        out, _ = run_hip(arch, sd, x, nz, G, o, prompt_lens=lens, eos_vec=eos)
    got, want = [h[0].seq.shape[0] for h in out.hypotheses], [h[0].seq.shape[0] for h in ref]
    assert got == want
    assert len(set(want)) > 1                                       # documents really finished at different steps
    worst, cos = compare(out.hypotheses, ref, sd)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)


def test_more_batch_sizes_than_the_graph_cache_holds():
    """24 documents with prompts of 1 ... 24 sentences finish one per step, so the session sees 24 different numbers of
    active documents and records a loop for each; it keeps 16, so the cache is dropped and refilled while the last replay
    may still be running (the executable graphs are destroyed only after that stream has drained).  On a non-default
    stream, against the oracle."""
    arch = "mini_two_tower_diffusion_lcm"
    B, G = 24, 2
    lens = list(range(1, B + 1))
    cfg, sd, x, nz = make_problem(arch, B, B, G, seed=4, prompt_lens=lens)
    opts = dict(DEFAULT_OPTS, inference_timesteps=5)
    ref = run_oracle(cfg, sd, x, nz, G, opts, prompt_lens=lens)
    stream = torch.cuda.Stream()
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        out, model = run_hip(arch, sd, x, nz, G, opts, prompt_lens=lens)
        again, _ = run_hip(arch, sd, x, nz, G, opts, prompt_lens=lens, model=model)
    assert [h[0].seq.shape[0] for h in out.hypotheses] == [n + G for n in lens]
    worst, cos = compare(out.hypotheses, ref, sd)
    assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (worst, cos)
    for ha, hb in zip(out.hypotheses, again.hypotheses):
        assert torch.equal(ha[0].seq, hb[0].seq)
    del model


def test_seed_option_and_device_drawn_noise():
    """Without `noise=` the draws come from torch.randn on the device, seeded by `options.seed` as the reference seeds
    them (generator.py:124-125): the same seed gives the same bits, another seed another generation, and a generation
    with the device's draws handed back in as `noise=` reproduces it."""
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.inference.two_tower_diffusion_lcm import TwoTowerDiffusionLCMGenerator
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    from parity_utils import mini_config
    arch = "mini_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 3, 3, 3)
    model = create_two_tower_diffusion_lcm_model(mini_config(), device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    batch = EmbeddingsBatch(x.cuda(), None)

    def run(seed):
        gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**dict(DEFAULT_OPTS, inference_timesteps=10, seed=seed)))
        return torch.stack([h[0].seq for h in gen(batch, max_gen_len=3).hypotheses])

    a, b, c = run(11), run(11), run(12)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all() and torch.equal(a[:, :3], x.cuda())
    # the same draws, made by hand in the order the generator makes them (one (B, 1, D) draw per sentence)
    torch.manual_seed(11)
    draws = torch.stack([torch.randn(3, 1, 1024, device="cuda:0")[:, 0] for _ in range(3)])
    gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**dict(DEFAULT_OPTS, inference_timesteps=10)))
    d = torch.stack([h[0].seq for h in gen(batch, max_gen_len=3, noise=draws).hypotheses])
    assert torch.equal(a, d)
    del model


def test_teacher_forced_scorer():
    """TwoTowerDiffusionLCMScorer against the oracle's restatement of the reference scorer: ragged documents,
    min_context_len 1 and 3 (with and without a prefill)."""
    from lcm.datasets.batch import EmbeddingsBatch, PaddingMask
    from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMScorer
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    from parity_utils import mini_config
    arch = "mini_two_tower_diffusion_lcm"
    lens = [6, 4, 5, 6]
    cfg, sd, x, nz = make_problem(arch, 4, 6, 0, prompt_lens=lens)
    opts = dict(DEFAULT_OPTS, inference_timesteps=5)
    model = create_two_tower_diffusion_lcm_model(mini_config(), device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    for mcl in (1, 3):
        ref = oracle_mod.OracleScorer(oracle_mod.OracleTwoTower(cfg, sd), oracle_mod.GenOptions(**opts))(
            x, text_lens=torch.tensor(lens), min_context_len=mcl, noise=nz)
        scorer = TwoTowerDiffusionLCMScorer(model, DiffusionLCMGeneratorOptions(**opts))
        out = scorer(EmbeddingsBatch(x.cuda(), PaddingMask(torch.tensor(lens, device="cuda:0"), 6)), min_context_len=mcl,
                     noise=nz.cuda())
        assert [h[0].seq.shape[0] for h in out.hypotheses] == [n - mcl for n in lens]
        worst, cos = compare(out.hypotheses, ref, sd)
        assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (mcl, worst, cos)


def test_packed_all_gather_through_rccl_single_rank():
    """The output exchange of `bench.py --gpus N` on device tensors through the `nccl` backend (RCCL): a one-rank
    process group on cuda:0 in a child process runs the same pack -> all_gather_into_tensor -> unpack path that N ranks
    run (with one rank RCCL's all-gather is a device copy, but backend set-up, the uint8 payload and the unpacking are
    the real ones).  More ranks need more GPUs than a test box has; the world-size-2 exchange is covered on gloo."""
    import os
    import subprocess
    import sys
    code = r"""
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ["PORT"], RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from lcm.utils.distributed import _gather_packed, all_gather_generations
seqs = torch.randn(5, 9, 1024, device="cuda").bfloat16()
lens = torch.tensor([9, 3, 7, 9, 1], device="cuda", dtype=torch.int32)
for width in (5, 8, None):
    out, olens = _gather_packed(seqs, lens, None, width)
    torch.cuda.synchronize()
    assert out.dtype == seqs.dtype and torch.equal(out, seqs) and torch.equal(olens, lens), width
same, _ = all_gather_generations(seqs, lens)      # one rank: nothing to exchange
assert same is seqs
dist.barrier()
dist.destroy_process_group()
print("rccl exchange ok")
"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PORT=str(port), PYTHONPATH=root)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "rccl exchange ok" in out.stdout, out.stderr[-3000:]
