# Importing the archs module registers the named architectures.
from lcm.models.two_tower_diffusion_lcm import archs as _archs  # noqa: F401
from lcm.models.two_tower_diffusion_lcm.builder import (  # noqa: F401
    TWO_TOWER_DIFFUSION_LCM_MODEL_TYPE,
    DenoiserConfig,
    EncoderFrontendConfig,
    TransformerConfig,
    TwoTowerDiffusionLCModel,
    TwoTowerDiffusionLCModelBuilder,
    TwoTowerDiffusionLCModelConfig,
    create_two_tower_diffusion_lcm_model,
    lcm_archs,
)
from lcm.models.two_tower_diffusion_lcm.loader import load_two_tower_diffusion_lcm_model  # noqa: F401
