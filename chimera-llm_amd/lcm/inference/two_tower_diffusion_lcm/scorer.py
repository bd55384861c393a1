"""Teacher-forced scoring with the two-tower diffusion LCM on the HIP path.

Same class name, constructor and call signature as the reference's
lcm/inference/two_tower_diffusion_lcm/scorer.py:27-314: every step feeds the gold previous sentence, records the
model's prediction for the next one (never feeding it back) and drops a document after its last sentence;
`hypotheses[i][0].seq` holds the predictions for positions min_context_len .. len_i - 1.  Built on the model hooks of
the generator (prefill / predict / reorder as slot-map updates); `noise=` supplies the N(0, 1) draws as
noise[step - min_context_len, document].
"""

import logging
from typing import List, Optional

import torch
from torch import Tensor

from lcm.datasets.batch import EmbeddingsBatch
from lcm.inference.output import GenerationCounters, Hypothesis, SequenceGeneratorOutput
from lcm.inference.two_tower_diffusion_lcm.generator import TwoTowerDiffusionLCMGenerator

logger = logging.getLogger(__name__)


class TwoTowerDiffusionLCMScorer(TwoTowerDiffusionLCMGenerator):
    """Score by generating in teacher-forcing mode with a two-tower diffusion LCM."""

    @torch.inference_mode()
    def __call__(self, batch_input: EmbeddingsBatch, max_gen_len: Optional[int] = None,       # type: ignore[override]
                 min_gen_len: Optional[int] = None, min_context_len: int = 1, temperature: float = 0.0,
                 disable_cache: bool = False, **kwargs) -> SequenceGeneratorOutput:
        opt, model = self.options, self.model
        if opt.seed:
            torch.manual_seed(opt.seed)
        noise: Optional[Tensor] = kwargs.get("noise")
        self.min_context_len = min_context_len
        seqs = batch_input.seqs
        B, T, D = seqs.shape
        device = seqs.device
        mask = batch_input.padding_mask
        if mask is None:
            text_lens = [T] * B
        else:
            assert mask.seq_lens is not None, "Expecting a valid `self.text_seq_lens` Tensor, found `None`"
            text_lens = [int(v) for v in mask.seq_lens.tolist()]
        self.max_text_len = T

        max_gen_len = min(max_gen_len or self.max_seq_len, T - min_context_len)
        assert max_gen_len <= self.max_seq_len, (
            f"Generator can generate up to {self.max_seq_len} sequences, max_gen_len={max_gen_len}")
        self.max_gen_len = max_gen_len
        min_gen_len = min_gen_len or self.min_seq_len
        assert min_gen_len is not None, "A `min_gen_len` is required"
        assert min_gen_len > 0, f"min_gen_len must be greater than or equal to 1, min_gen_len={min_gen_len}"
        self.min_gen_len = min_gen_len
        self.temperature = temperature if temperature != 0.0 else opt.lcm_temperature

        self._new_state(T, disable_cache)
        self.seqs = seqs
        self.preds = torch.zeros((B, T - min_context_len, D), device=device, dtype=model.dtype)
        self.step_scores = torch.zeros((B, T), device=device)
        self.hypotheses: List[List[Hypothesis]] = [[] for _ in range(B)]
        active = list(range(B))
        active_t = torch.arange(B, device=device)
        try:
            self.step_nr = min_context_len
            prefill_len = min_context_len - 1
            if prefill_len > 0:
                self._prefill(self.seqs[:, :prefill_len])
            else:
                logger.warning("Skipping prefill since only a context size of %d is provided in the prefix", min_context_len)
            for step in range(min_context_len, T):
                self.step_nr = step
                everyone = len(active) == B
                prev = self.seqs[:, step - 1:step] if everyone else self.seqs[active_t, step - 1:step]
                nz = None
                if noise is not None:
                    nz = noise[step - min_context_len].reshape(B, -1)
                    nz = nz if everyone else nz[active_t]
                new = self._predict(prev, nz, active_t, everyone)                # (n_act, D) fp32
                if everyone:
                    self.preds[:, step - min_context_len] = new.to(self.preds.dtype)
                else:
                    self.preds[active_t, step - min_context_len] = new.to(self.preds.dtype)
                keep = [r for r, d in enumerate(active) if text_lens[d] > step + 1]
                if len(keep) != len(active):
                    for r, d in enumerate(active):
                        if text_lens[d] <= step + 1:
                            self.hypotheses[d] = [Hypothesis(seq=self.preds[d, :text_lens[d] - min_context_len], score=None,
                                                             step_scores=self.step_scores[d])]
                    if not keep:
                        break
                    active = [active[r] for r in keep]
                    active_t = torch.tensor(active, device=device)
                    self.reorder_state(keep)
        finally:
            self._release()
        return SequenceGeneratorOutput(self.hypotheses, counters=GenerationCounters())
