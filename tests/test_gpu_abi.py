"""The export / header checks of test_abi.py again under the `gpu` marker, so that they also run against the library
as it arrives on the GPU box (the CPU run covers the container's build)."""

import pytest
import torch

import test_abi

pytestmark = pytest.mark.gpu

hip = test_abi.hip


def test_every_declared_symbol_is_exported_on_the_gpu_box(hip):
    test_abi.test_every_declared_symbol_is_exported(hip)


def test_error_channel_and_model_table_validation_on_the_gpu_box(hip):
    test_abi.test_version_and_error_channel(hip)
    test_abi.test_model_table_validation_without_gpu(hip)


def test_session_profile_over_many_launches():
    """Per-launch profiling keeps a bounded number of HIP event pairs outstanding (the stream is drained and the
    pairs collected every 1024 launches): a profiled sentence of 100 diffusion steps (3000 launches) must still count
    every launch and time every family."""
    import oracle
    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    from parity_utils import DEFAULT_OPTS, make_problem, mini_config
    cfg, sd, x, nz = make_problem("mini_two_tower_diffusion_lcm", 3, 4, 1)
    model = create_two_tower_diffusion_lcm_model(mini_config(), device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**dict(DEFAULT_OPTS, inference_timesteps=100)))
    batch = EmbeddingsBatch(x.cuda(), None)
    plain = gen(batch, max_gen_len=1, noise=nz.cuda())
    sess = model._session_pool[0]
    sess.profile_enable(True)
    profiled = gen(batch, max_gen_len=1, noise=nz.cuda())
    rows = {r["name"]: r for r in sess.profile()}
    sess.profile_enable(False)
    steps, layers = 100, 3
    assert rows["den.norm_mod"]["launches"] == steps * (3 * layers + 1)
    assert rows["den.cross_attn"]["launches"] == steps * layers
    assert rows["den.ddim_cfg_step"]["launches"] == steps
    assert sum(r["launches"] for r in rows.values()) > 2048
    assert all(r["total_ms"] > 0 for r in rows.values() if r["launches"])
    for a, b in zip(plain.hypotheses, profiled.hypotheses):
        assert torch.equal(a[0].seq, b[0].seq)
