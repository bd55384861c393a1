"""Model-card + checkpoint loading on the HIP path, the way the reference's cards are written
(lcm/utils/card_utils.py:178-223, README.md:187-199): `checkpoint: file://.../model.pt` holding
`{"model": state_dict}`, DDP `module.` prefixes (lcm/models/base_lcm/loader.py:24-30), half-precision
checkpoints cast to the requested dtype, `model_config` overrides, and the normalizer taken from its own card when
the checkpoint does not carry it."""

import os
import pytest
import torch
import yaml

import oracle
from parity_utils import DEFAULT_OPTS, compare, make_problem, run_hip, run_oracle

pytestmark = pytest.mark.gpu


def test_two_tower_card_roundtrip(tmp_path):
    from lcm.utils.card_utils import load_model_from_card
    arch = "toy_two_tower_diffusion_lcm"
    cfg, sd, x, nz = make_problem(arch, 2, 3, 2, seed=5)
    # a checkpoint as a DDP fp16 training run leaves it, without the normalizer buffers
    ckpt = {"model": {"module." + k: v.half() for k, v in sd.items() if not k.startswith("sonar_normalizer.")}}
    torch.save(ckpt, tmp_path / "model.pt")
    card = {"name": "tmp_two_tower", "model_family": "two_tower_diffusion_lcm", "model_arch": arch,
            "checkpoint": "file://" + str(tmp_path / "model.pt"),
            "model_config": {"trained_with_cf_guidance": True, "noise_scheduler": {"num_diffusion_train_steps": 1000}}}
    with open(tmp_path / "model_card.yaml", "w") as f:
        yaml.safe_dump(card, f)
    model = load_model_from_card(str(tmp_path / "model_card.yaml"), device="cuda:0", dtype=torch.float32)
    assert model.config.trained_with_cf_guidance and model.dtype == torch.float32
    # the dummy normalizer card supplied center / scale
    assert torch.allclose(model.sonar_normalizer.center.cpu(), sd["sonar_normalizer.center"])
    sd16 = {k: (v if k.startswith("sonar_normalizer.") else v.half().float()) for k, v in sd.items()}
    opts = dict(DEFAULT_OPTS, inference_timesteps=5)
    ref = run_oracle(cfg, sd16, x, nz, 2, opts)
    out, _ = run_hip(arch, None, x, nz, 2, opts, model=model)
    worst, cos = compare(out.hypotheses, ref, sd, start=3)
    assert worst <= 1e-3 and cos >= 0.9999, (worst, cos)


def test_base_card_roundtrip_and_directory_card(tmp_path):
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.utils.card_utils import load_model_from_card
    cfg = oracle.get_base_arch("toy_base_lcm")
    sd = oracle.init_state_dict(cfg, seed=9, qk_gain=2.0, pos_std=0.5)
    torch.save({"model": sd}, tmp_path / "model.pt")
    with open(tmp_path / "model_card.yaml", "w") as f:
        yaml.safe_dump({"name": "tmp_base", "model_family": "base_lcm", "model_arch": "toy_base_lcm",
                        "checkpoint": "model.pt"}, f)          # path relative to the card, card given as a directory
    model = load_model_from_card(str(tmp_path), device="cuda:0", dtype=torch.float32)
    x = torch.randn(2, 5, 1024)
    ref = oracle.OracleBaseLCM(cfg, sd).predict(x)
    out = model.predict_next_sentence(EmbeddingsBatch(x.cuda(), None)).seqs.cpu()
    assert (out - ref).abs().max() <= 2e-4 * ref.abs().max()


def test_strict_loading_errors():
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs
    model = create_two_tower_diffusion_lcm_model(lcm_archs.get_config("toy_two_tower_diffusion_lcm"), device="cuda:0")
    sd = dict(model.state_dict())
    bad = dict(sd)
    bad.pop("denoiser.final_proj.fc.bias")
    with pytest.raises(RuntimeError, match="missing keys"):
        model.load_state_dict(bad)
    bad = dict(sd)
    bad["denoiser.final_proj.fc.bias"] = torch.zeros(7)
    with pytest.raises(RuntimeError, match="shape mismatch"):
        model.load_state_dict(bad)


def test_fresh_models_carry_their_normalizer_card_statistics():
    """abstract_lcm/builder.py:89-101: a model is built around the normalizer its config names, loaded from that card, so
    `create_*_model` / the builder classes return a model that already normalizes with the fitted statistics.  (The
    seeded `init_parameters` of this package resets every tensor, the statistics included.)  The normalizer's
    FFTInterface methods are the packed transform of the "*_fft" archs."""
    from lcm.models.base_lcm import BaseLCModelBuilder, lcm_archs as base_archs
    from lcm.models.two_tower_diffusion_lcm import TwoTowerDiffusionLCModelBuilder, create_two_tower_diffusion_lcm_model, lcm_archs
    dummy = oracle.load_dummy_normalizer()
    cfg = lcm_archs.get_config("toy_two_tower_diffusion_lcm")
    for model in (create_two_tower_diffusion_lcm_model(cfg, device="cuda:0"),
                  TwoTowerDiffusionLCModelBuilder(cfg, device="cuda:0", dtype=torch.float32).build_model()):
        assert torch.equal(model.sonar_normalizer.center.cpu(), dummy["center"])
        assert torch.equal(model.state_dict()["sonar_normalizer.scale"].cpu(), dummy["scale"])
        x = torch.randn(3, 1024, device="cuda:0") * 0.01
        assert torch.allclose(model.sonar_normalizer.normalize(x).cpu(), (x.cpu() - dummy["center"]) / dummy["scale"], atol=1e-4)
        model.init_parameters(seed=1)
        assert float(model.sonar_normalizer.center.abs().max()) == 0.0
    bcfg = base_archs.get_config("toy_base_lcm")
    bcfg.sonar_normalizer_name = "dummy_sonar_normalizer"
    base = BaseLCModelBuilder(bcfg, device="cuda:0", dtype=torch.float32).build_model()
    assert torch.equal(base.sonar_normalizer.scale.cpu(), dummy["scale"])
    n = base.sonar_normalizer
    v = torch.randn(2, 1024, device="cuda:0")
    assert (n.fft_inverse_transform(n.fft_transform(v)) - v).abs().max() <= 1e-5
    assert (n.fft_transform(v).cpu() - oracle.model.fft_transform(v.cpu())).abs().max() <= 1e-4


def test_training_directory_and_checkpoint_path_as_model_names(tmp_path):
    """`load_model_from_card` given a training directory (last checkpoint) or a `checkpoints/step_<n>/model.pt` inside one,
    and `load_model_with_overrides` for a directory with a card beside the checkpoint and for an old-style directory with
    `config_logs/all_config.yaml` (card_utils.py:72-125, 178-223): the weights of the right step arrive in the model."""
    from lcm.utils.card_utils import load_model_from_card, load_model_with_overrides
    arch = "toy_base_lcm"
    cfg = oracle.get_base_arch(arch)
    sds = {}
    for step in (10, 20):
        sds[step] = oracle.init_state_dict(cfg, seed=step)
        d = tmp_path / "run" / "checkpoints" / f"step_{step}"
        d.mkdir(parents=True)
        torch.save({"model": {"module." + k: v for k, v in sds[step].items()}}, d / "model.pt")
        yaml.safe_dump({"name": f"run_step{step}", "model_family": "base_lcm", "model_arch": arch,
                        "checkpoint": "file://" + str(d / "model.pt")}, open(d / "model_card.yaml", "w"))
    key = "lcm.layers.1.ffn.inner_proj.weight"
    m = load_model_from_card(str(tmp_path / "run"), device="cuda:0", dtype=torch.float32)
    assert torch.equal(m.state_dict()[key].cpu(), sds[20][key])
    m = load_model_from_card(str(tmp_path / "run" / "checkpoints" / "step_10" / "model.pt"), device="cuda:0", dtype=torch.float32)
    assert torch.equal(m.state_dict()[key].cpu(), sds[10][key])
    m = load_model_with_overrides(tmp_path / "run", step=10, device="cuda:0", dtype=torch.float32)
    assert torch.equal(m.state_dict()[key].cpu(), sds[10][key])
    old = tmp_path / "old_run"
    (old / "config_logs").mkdir(parents=True)
    torch.save({"model": sds[20]}, old / "model.pt")
    yaml.safe_dump({"trainer": {"model_config_or_name": {"lcm": {"num_layers": 2}}}}, open(old / "config_logs" / "all_config.yaml", "w"))
    m = load_model_with_overrides(old, model_type="base_lcm", device="cuda:0", dtype=torch.float32)
    assert torch.equal(m.state_dict()[key].cpu(), sds[20][key])
    with pytest.raises(ValueError, match="not a valid model directory"):
        load_model_with_overrides(tmp_path / "nothing_here", model_type="base_lcm")


def test_reference_sonar_normalizer_loads_by_card_name():
    """The reference's own test (tests/units/models/test_lcm_architecture.py:34-35), call for call, plus what the card
    must deliver: the fixture's statistics on the device, and a normalize / denormalize round trip through them."""
    from lcm.models.sonar_normalizer import SonarNormalizerConfig, create_sonar_normalizer
    from lcm.models.sonar_normalizer.loader import load_sonar_normalizer_model
    model = load_sonar_normalizer_model("dummy_sonar_normalizer", device="cuda:0", dtype=torch.float32)
    assert model is not None
    blob = torch.load(os.path.join(os.path.dirname(__file__), "golden", "dummy_normalizer.pt"), map_location="cpu", weights_only=True)
    blob = blob["model"] if "model" in blob else blob
    assert torch.equal(model.center.cpu(), blob["center"].float()) and torch.equal(model.scale.cpu(), blob["scale"].float())
    x = torch.randn(7, 1024, generator=torch.Generator().manual_seed(1)).cuda() * 0.01
    assert float((model.denormalize(model.normalize(x)) - x).abs().max()) <= 1e-6
    fresh = create_sonar_normalizer(SonarNormalizerConfig(dim=1024), device="cuda:0", dtype=torch.float32)
    assert fresh.center.shape == (1024,) and fresh.config.dim == 1024
