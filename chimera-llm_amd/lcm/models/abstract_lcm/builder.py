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
