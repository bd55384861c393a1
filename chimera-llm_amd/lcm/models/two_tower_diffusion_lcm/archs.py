"""Named architectures (values of the reference's lcm/models/two_tower_diffusion_lcm/archs.py:17-112)."""

from lcm.models.two_tower_diffusion_lcm.builder import (
    DenoiserConfig,
    TransformerConfig,
    TwoTowerDiffusionLCModelConfig,
    lcm_arch,
)
from lcm.nn.schedulers import DDIMSchedulerConfig


@lcm_arch("toy_two_tower_diffusion_lcm")
def toy_lcm() -> TwoTowerDiffusionLCModelConfig:
    return TwoTowerDiffusionLCModelConfig(
        context_encoder=TransformerConfig(num_layers=2),
        denoiser=DenoiserConfig(num_layers=2),
        sonar_normalizer_name="dummy_sonar_normalizer",
    )


def _scaled(model_dim: int, heads: int, denoiser_layers: int) -> TwoTowerDiffusionLCModelConfig:
    shared = dict(ffn_inner_dim=4 * model_dim, num_attn_heads=heads, final_dropout_p=0.0, attention_dropout_p=0.0,
                  dropout_p=0.1, mha_output_proj_bias=True, use_swiglu=True, layer_normalization_style="rms")
    return TwoTowerDiffusionLCModelConfig(
        model_dim=model_dim,
        max_seq_len=4096,
        context_encoder=TransformerConfig(num_layers=5, pos_embedding_style="rope", **shared),
        denoiser=DenoiserConfig(num_layers=denoiser_layers, timestep_embed_dim=model_dim,
                                pos_embedding_style="none", **shared),
        sonar_normalizer_name="dummy_sonar_normalizer",
        trained_with_cf_guidance=True,
        noise_scheduler=DDIMSchedulerConfig(num_diffusion_train_steps=100),
    )


@lcm_arch("two_tower_diffusion_lcm_1_6B")
def two_tower_diffusion_lcm_1_6B() -> TwoTowerDiffusionLCModelConfig:
    """5-layer contextualizer / 13-layer denoiser / model dim 2048."""
    return _scaled(2048, 16, 13)


@lcm_arch("two_tower_diffusion_lcm_7B")
def two_tower_diffusion_lcm_7B() -> TwoTowerDiffusionLCModelConfig:
    """5-layer contextualizer / 14-layer denoiser / model dim 4096 (6,930,781,696 parameters)."""
    return _scaled(4096, 32, 14)
