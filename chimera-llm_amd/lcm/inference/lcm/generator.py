"""Options and constructor shared by the LCM generators.

Field names and defaults follow the reference's lcm/inference/lcm/generator.py:29-58; the
constructor checks follow :64-97.  Only the two-tower diffusion generator runs on the HIP path.
"""

import logging
from dataclasses import dataclass
from typing import Optional

import torch

logger = logging.getLogger(__name__)


@dataclass
class LCMGeneratorOptions:
    max_seq_len: int = 200                 # hard limit on generated length
    min_seq_len: int = 1
    eos_threshold: Optional[float] = 0.9   # cosine similarity to the EOS vector that stops a document
    sample_latent_variable: bool = True
    stop_on_repetition_cosine_threshold: Optional[float] = None
    include_eos_token: bool = False
    trim_hypotheses: bool = False
    seed: Optional[int] = None
    lcm_temperature: float = 1.0


class LCMGenerator:
    def __init__(self, model, options: Optional[LCMGeneratorOptions] = None,
                 eos_vec: Optional[torch.Tensor] = None) -> None:
        model.eval()
        self.model = model
        options = options if options is not None else LCMGeneratorOptions()
        self.options = options
        self.eos_vec = eos_vec
        if eos_vec is None and options.eos_threshold:
            logger.warning("eos_threshold is set to %s, but eos_vec is not provided", options.eos_threshold)
        self.max_seq_len = options.max_seq_len
        self.min_seq_len = options.min_seq_len
        assert self.min_seq_len >= 1, (
            f"min_seq_len must be greater than or equal to 1, min_seq_len={options.min_seq_len}")
        self.eos_threshold = options.eos_threshold
        self.step_nr = 0
