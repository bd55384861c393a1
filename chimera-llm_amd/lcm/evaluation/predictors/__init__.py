"""Predictor registry of the LCM families (the part of lcm/evaluation/predictors/__init__.py:15-58 that names them)."""

from lcm.evaluation.predictors.lcm import LCMConfig, LCMPredictor
from lcm.evaluation.predictors.two_tower_diffusion_lcm import TwoTowerDiffusionLCMConfig, TwoTowerDiffusionLCMPredictor

_PREDICTOR_CONFIG_MAP = {"base_lcm": LCMConfig, "two_tower_diffusion_lcm": TwoTowerDiffusionLCMConfig}


def get_config_cls(name: str):
    if name not in _PREDICTOR_CONFIG_MAP:
        raise ValueError(f"No predictor registered under the name {name}")
    return _PREDICTOR_CONFIG_MAP[name]


def build_predictor(config, **kwargs):
    return config.predictor_class().from_config(config, **kwargs)


__all__ = ["LCMConfig", "LCMPredictor", "TwoTowerDiffusionLCMConfig", "TwoTowerDiffusionLCMPredictor",
           "get_config_cls", "build_predictor"]
