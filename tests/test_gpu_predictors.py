"""The evaluation-side callers of the generators: `lcm.evaluation.predictors` (reference
lcm/evaluation/predictors/lcm.py:33-269, two_tower_diffusion_lcm.py:18-47) with stand-ins for the SONAR text pipelines."""

import pytest
import torch

from parity_utils import DEFAULT_OPTS, base_config, mini_config

pytestmark = pytest.mark.gpu


class FakeDecoder:
    """Stands for EmbeddingToTextModelPipeline: one string per embedding row, and a record of the call."""

    def __init__(self):
        self.calls = []

    def predict(self, embeddings, target_lang, max_seq_len, temperature):
        self.calls.append((tuple(embeddings.shape), target_lang, max_seq_len, temperature))
        return [f"sentence {i}: {float(e.float().sum()):.6f}" for i, e in enumerate(embeddings)]


class FakeEncoder:
    """Stands for TextToEmbeddingModelPipeline: a deterministic embedding per sentence."""

    def predict(self, sentences, source_lang):
        assert source_lang == "eng_Latn"
        rows = []
        for s in sentences:
            g = torch.Generator().manual_seed(sum(s.encode()))
            rows.append(torch.randn(1024, generator=g) * 0.006)
        return torch.stack(rows)


def test_two_tower_predictor_matches_the_generator_it_wraps(tmp_path):
    from lcm.evaluation.api import EOSConfig
    from lcm.evaluation.predictors import TwoTowerDiffusionLCMConfig, TwoTowerDiffusionLCMPredictor, build_predictor
    from lcm.inference.predict import generate_continuations
    from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model

    model = create_two_tower_diffusion_lcm_model(mini_config(), device="cuda:0", dtype=torch.float32, seed=0)
    model.init_parameters(seed=5, modulator_std=0.2)
    opts = dict(DEFAULT_OPTS, inference_timesteps=6, seed=21)
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randn(n, 1024, generator=g) * 0.006 for n in (4, 2, 3, 1, 2)]

    config = TwoTowerDiffusionLCMConfig(model_card="unused", generator_batch_size=2, **opts)
    assert config.predictor_class() is TwoTowerDiffusionLCMPredictor
    decoder = FakeDecoder()
    predictor = build_predictor(config, model=model, device="cuda:0", dtype="torch.float32", text_decoder=decoder)
    assert isinstance(predictor, TwoTowerDiffusionLCMPredictor) and isinstance(predictor.generator, TwoTowerDiffusionLCMGenerator)
    preds = predictor(prompts, max_prompt_len=3, max_gen_len=2)

    direct = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**opts))
    want = generate_continuations(direct, [p.cuda() for p in prompts], max_gen_len=2, max_prompt_len=3, generator_batch_size=2)
    assert len(preds) == 5
    for p, w in zip(preds, want):
        assert torch.equal(p.embed, w) and p.embed.shape == (2, 1024) and p.logprobs is None
    # every document's sentences went through the decoder in ONE call, and came back to their documents in order
    assert decoder.calls == [((10, 1024), "eng_Latn", 256, 1.0)]
    flat = decoder.predict(torch.cat(want), "eng_Latn", 256, 1.0)
    assert [p.text for p in preds] == [flat[2 * i:2 * i + 2] for i in range(5)]

    # without a decoder: embeddings only; the run's temperature and options land in the config
    bare = TwoTowerDiffusionLCMPredictor(config, model=model, device="cuda:0", dtype=torch.float32, temperature=0.5)
    assert bare.config.lcm_temperature == 0.5
    out = bare(prompts[:2], max_gen_len=1)
    assert [p.text for p in out] == [None, None] and out[0].embed.shape == (1, 1024)

    # text prompts: split, encoded sentence by sentence, then the same path; refused without an encoder
    enc = FakeEncoder()
    texted = TwoTowerDiffusionLCMPredictor(config, model=model, device="cuda:0", dtype=torch.float32, text_encoder=enc,
                                           split_text=lambda t: t.split(". "))
    texts = ["One. Two. Three", "Four"]
    got = texted(texts, max_gen_len=1)
    want = generate_continuations(direct, [enc.predict(t.split(". "), "eng_Latn").cuda() for t in texts], max_gen_len=1,
                                  generator_batch_size=2)
    assert all(torch.equal(a.embed, b) for a, b in zip(got, want))
    with pytest.raises(RuntimeError, match="text_encoder"):
        bare(texts, max_gen_len=1)
    with pytest.raises(AssertionError, match="greedy"):
        bare(prompts[:1], greedy=False)

    # an EOS vector from a saved tensor: generation stops where the generator with that vector stops
    eos = want[0][0].float().cpu()
    torch.save(eos, tmp_path / "eos.pt")
    stop_cfg = TwoTowerDiffusionLCMConfig(model_card="unused", **dict(opts, eos_threshold=0.9))
    stopper = TwoTowerDiffusionLCMPredictor(stop_cfg, model=model, device="cuda:0", dtype=torch.float32,
                                            eos_config=EOSConfig(ckpt=str(tmp_path / "eos.pt")))
    assert torch.equal(stopper.eos_vec.cpu(), eos)
    ref_gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**dict(opts, eos_threshold=0.9)), eos_vec=eos.cuda())
    a = stopper([enc.predict(texts[0].split(". "), "eng_Latn")], max_gen_len=3)
    b = generate_continuations(ref_gen, [enc.predict(texts[0].split(". "), "eng_Latn").cuda()], max_gen_len=3)
    assert torch.equal(a[0].embed, b[0])


def test_base_predictor_and_registry():
    from lcm.evaluation.predictors import LCMConfig, LCMPredictor, TwoTowerDiffusionLCMConfig, get_config_cls
    from lcm.inference.lcm import LCMGenerator, LCMGeneratorOptions
    from lcm.inference.predict import generate_continuations
    from lcm.models.base_lcm import create_base_lcm_model

    assert get_config_cls("base_lcm") is LCMConfig and get_config_cls("two_tower_diffusion_lcm") is TwoTowerDiffusionLCMConfig
    with pytest.raises(ValueError, match="No predictor registered"):
        get_config_cls("llama3")
    model = create_base_lcm_model(base_config("toy_base_lcm"), device="cuda:0", dtype=torch.bfloat16, seed=0)
    config = LCMConfig(model_card="unused", max_seq_len=3, seed=4)        # the base model samples: same seed, same draws
    predictor = LCMPredictor.from_config(config, model=model, device="cuda:0", dtype="torch.float16")
    assert predictor.dtype == torch.bfloat16                       # the reference's evaluation default is served in bf16
    g = torch.Generator().manual_seed(8)
    prompts = [torch.randn(n, 1024, generator=g) * 0.006 for n in (3, 5)]
    preds = predictor(prompts)                                      # max_gen_len from the config
    want = generate_continuations(LCMGenerator(model, LCMGeneratorOptions(max_seq_len=3, seed=4)), [p.cuda() for p in prompts], max_gen_len=3)
    assert [tuple(p.embed.shape) for p in preds] == [(3, 1024), (3, 1024)]
    assert all(torch.equal(p.embed, w) for p, w in zip(preds, want))
