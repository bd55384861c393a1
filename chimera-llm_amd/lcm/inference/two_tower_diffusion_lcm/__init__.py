from lcm.inference.two_tower_diffusion_lcm.generator import (  # noqa: F401
    DiffusionLCMGeneratorOptions,
    TwoTowerDiffusionLCMGenerator,
)
from lcm.inference.two_tower_diffusion_lcm.scorer import TwoTowerDiffusionLCMScorer  # noqa: F401,E402
