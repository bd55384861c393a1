"""Result containers of the generators (the reference takes them from fairseq2.generation)."""

from dataclasses import dataclass, field
from typing import List, Optional

from torch import Tensor


@dataclass
class Hypothesis:
    seq: Tensor                      # (len, D): prompt followed by the generated embeddings
    score: Optional[Tensor] = None
    step_scores: Optional[Tensor] = None


@dataclass
class GenerationCounters:
    prefill_size: int = 0
    num_elements: int = 0
    generation_time: float = 0.0
    cache_size: int = 0
    cache_capacity: int = 0


@dataclass
class SequenceGeneratorOutput:
    hypotheses: List[List[Hypothesis]]
    counters: GenerationCounters = field(default_factory=GenerationCounters)
