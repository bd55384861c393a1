"""Base config / model types (reference: lcm/models/abstract_lcm/builder.py:25-106)."""

from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class AbstractLCModelConfig:
    model_type: str = "abstract_lcm"
    sonar_embed_dim: int = 1024
    sonar_normalizer_name: Optional[str] = None


class AbstractLCModel:
    """A model whose parameters are device tensors bound into a HIP weight table."""

    def __init__(self, config: AbstractLCModelConfig) -> None:
        self.config = config
        self.training = False

    @property
    def dtype(self) -> torch.dtype:
        return next(iter(self.state_dict().values())).dtype

    @property
    def device(self) -> torch.device:
        return next(iter(self.state_dict().values())).device

    def eval(self):
        self.training = False
        return self

    def state_dict(self):
        raise NotImplementedError


class AbstractLCModelBuilder:
    """Builds an LCM from its config (reference: abstract_lcm/builder.py:60-106)."""

    def __init__(self, config: AbstractLCModelConfig, *, device=None, dtype: Optional[torch.dtype] = None) -> None:
        self.config = config
        self.device, self.dtype = device, dtype

    def build_sonar_normalizer(self):
        """The normalizer the config names, loaded from its card; None without a name."""
        if self.config.sonar_normalizer_name is None:
            return None
        from lcm.models.sonar_normalizer.loader import load_sonar_normalizer_model
        return load_sonar_normalizer_model(self.config.sonar_normalizer_name, device=self.device, dtype=self.dtype)

    def build_model(self) -> AbstractLCModel:
        raise NotImplementedError
