"""Sentence-by-sentence generation with the two-tower diffusion LCM on the HIP path.

Same constructor, call signature, option names and output type as the reference's
lcm/inference/two_tower_diffusion_lcm/generator.py:29-213.  The generation loop and the stopping rules
(`finalize_step`, :305-423) are the shared ones of lcm/inference/lcm/generator.py in this package; this class
supplies the model hooks: prefill through `encode` (:222-259), one `predict_next_sentence` per step (:261-303)
and `reorder_state` (:442-466) as an update of the session's row -> cache-slot map for both state bags.
"""

import logging
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import Tensor

from lcm.datasets.batch import EmbeddingsBatch
from lcm.inference.lcm.generator import LCMGenerator, LCMGeneratorOptions
from lcm.inference.output import GenerationCounters, Hypothesis, SequenceGeneratorOutput
from lcm.models.two_tower_diffusion_lcm.builder import TwoTowerDiffusionLCModel
from lcm.nn.incremental_state import LCMIncrementalStateBag

logger = logging.getLogger(__name__)


@dataclass
class DiffusionLCMGeneratorOptions(LCMGeneratorOptions):
    guidance_scale: float = 1.0           # w of classifier-free guidance; 1 = no guidance
    guidance_rescale: float = 0.0         # phi of CFG-with-rescale (arXiv 2305.08891, alg. 2)
    ddim_eta: float = 0.0                 # 0 = deterministic DDIM; > 0 adds eta * sigma_t * N(0, 1) at every step
    epsilon_scaling: Optional[float] = None
    initial_noise_scale: float = 1.0
    inference_timesteps: int = 100
    clip_noise: int = 100
    thresholding: bool = False
    dynamic_thresholding_ratio: float = 0.995
    sample_max_value: float = 6.0


class TwoTowerDiffusionLCMGenerator(LCMGenerator):
    options: DiffusionLCMGeneratorOptions

    def __init__(self, model, options: Optional[LCMGeneratorOptions] = None,
                 eos_vec: Optional[Tensor] = None) -> None:
        super().__init__(model, options, eos_vec)
        assert isinstance(self.model, TwoTowerDiffusionLCModel), (
            "The TwoTowerDiffusionLCMGenerator expects a Diffusion LCM")
        logger.info("Setting up the model with decoding_options: %s -- %s", self.options, type(self.options))
        model.prep_for_denoising(self.options)

    takes_option_overrides = False     # unlike the base generator, extra kwargs do not override options (reference :98-213)

    # ---- model hooks of the shared loop (lcm/inference/lcm/generator.py) ---------------------------------------
    def _new_state(self, total: int, disable_cache: bool) -> None:
        if disable_cache:
            # the reference asserts a context state bag in prefill() (generator.py:226-228)
            raise AssertionError("Expecting a context state bag to prefill")
        self.state_bag = LCMIncrementalStateBag(total)
        self.context_state_bag = LCMIncrementalStateBag(total)
        self.context = None

    def _prefill(self, prompt: Tensor) -> None:
        """reference generator.py:222-259"""
        self.context = self.model.encode(EmbeddingsBatch(prompt, None), state_bag=self.context_state_bag, normalized=False)
        self.context_state_bag.increment_step_nr(prompt.shape[1])

    def _predict(self, prev: Tensor, noise: Optional[Tensor], active_t: Tensor, everyone: bool) -> Tensor:
        """reference generator.py:261-303"""
        out, self.context = self.model.predict_next_sentence(
            EmbeddingsBatch(prev, None), self.context, temperature=self.temperature,
            state_bag=self.state_bag, context_state_bag=self.context_state_bag, noise=noise,
            variance_noise=getattr(self, "_variance_step", None))
        self.state_bag.increment_step_nr()
        return out.seqs[:, -1]

    def _release(self) -> None:
        self.model.release_session(self.context_state_bag.session)
        self.context_state_bag.session = None

    def reorder_state(self, new_order) -> None:
        """Drops finished rows from the device state: `new_order` are the surviving row indices."""
        self.state_bag.reorder(new_order)
        self.context_state_bag.reorder(new_order)
