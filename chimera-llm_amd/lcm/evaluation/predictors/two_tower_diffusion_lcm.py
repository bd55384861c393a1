"""`TwoTowerDiffusionLCMConfig` / `TwoTowerDiffusionLCMPredictor` (reference
lcm/evaluation/predictors/two_tower_diffusion_lcm.py:18-47): the base predictor with the diffusion generator and its
option set."""

from dataclasses import dataclass
from typing import Any

from lcm.evaluation.predictors.lcm import LCMConfig, LCMPredictor
from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator


@dataclass(unsafe_hash=True)
class TwoTowerDiffusionLCMConfig(DiffusionLCMGeneratorOptions, LCMConfig):
    @classmethod
    def predictor_class(cls):
        return TwoTowerDiffusionLCMPredictor


class TwoTowerDiffusionLCMPredictor(LCMPredictor):
    config: TwoTowerDiffusionLCMConfig

    def build_generator(self, model) -> None:
        self.generator = TwoTowerDiffusionLCMGenerator(model, self.config, eos_vec=self.eos_vec)

    @staticmethod
    def from_config(config: LCMConfig, **kwargs: Any) -> "TwoTowerDiffusionLCMPredictor":
        return TwoTowerDiffusionLCMPredictor(config=config, **kwargs)
