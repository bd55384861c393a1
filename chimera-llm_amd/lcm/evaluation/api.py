"""The predictor-side types of the reference's lcm/evaluation/api.py that the LCM predictors use."""

import abc
from dataclasses import dataclass
from typing import List, Optional, Union

import torch


@dataclass
class Prediction:
    """One document's continuation (reference api.py:82-89)."""
    text: Union[str, List[str], None]
    embed: Optional[torch.Tensor] = None
    tokens: Optional[Union[List[int], List[str]]] = None
    logprobs: Optional[List[float]] = None
    text_offsets: Optional[List[int]] = None
    token_ids: Optional[List[int]] = None


@dataclass
class EOSConfig:
    """How the (untrained) end-of-text sentence is represented: a text to encode or a saved vector (api.py:333-345)."""
    text: Optional[str] = None
    ckpt: Optional[str] = None


class PredictorConfig:
    @classmethod
    @abc.abstractmethod
    def predictor_class(cls):
        ...
