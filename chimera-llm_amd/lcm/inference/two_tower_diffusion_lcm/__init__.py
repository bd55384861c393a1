from lcm.inference.two_tower_diffusion_lcm.generator import (  # noqa: F401
    DiffusionLCMGeneratorOptions,
    TwoTowerDiffusionLCMGenerator,
)
