"""Teacher-forced scoring with an LCM on the HIP path.

Same class name, constructor and call signature as the reference's lcm/inference/lcm/scorer.py:19-198: every step
feeds the gold previous sentence, records the model's prediction for the next one (never feeding it back) and drops
a document after its last sentence; `hypotheses[i][0].seq` holds the predictions for positions
min_context_len .. len_i - 1 of document i.

`LCMScorer` drives a base LCM.  A base LCM's prediction for position j depends only on the gold positions before j
(causal decoder, no sampling loop), so the default path scores a batch in ONE pass over all positions
(`lcmhip_base_forward` with all_positions: many-row matrix-core GEMMs and the prefill attention kernel) instead of one
decode launch chain per sentence; the step-by-step loop of the reference (:118-133, with the KV cache or, with
`disable_cache`, re-encoding the prefix) is kept as `one_pass = False` and gives the same predictions.  The two-tower
diffusion scorer has a sampling loop per sentence and always runs step by step: it inherits `_score_step_by_step` from
here, as the reference's scorers share the loop through their generators.

Extension: `noise=` supplies the N(0, 1) draws as noise[step - min_context_len, document].
"""

import logging
from typing import List, Optional

import torch
from torch import Tensor

from lcm.datasets.batch import EmbeddingsBatch
from lcm.inference.lcm.generator import LCMGenerator
from lcm.inference.output import GenerationCounters, Hypothesis, SequenceGeneratorOutput

logger = logging.getLogger(__name__)


class LCMScorer(LCMGenerator):
    """Generates with an LCM model in teacher-forcing mode."""

    one_pass = True        # base LCM: all positions in one forward; False: the reference's step-by-step loop

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
        mask = batch_input.padding_mask
        if mask is None:
            text_lens = [T] * B
        else:
            assert mask.seq_lens is not None, "Expecting a valid `self.text_seq_lens` Tensor, found `None`"
            text_lens = [int(v) for v in mask.seq_lens.tolist()]
        self.max_text_len = T

        max_gen_len = min(max_gen_len or self.max_seq_len, T - min_context_len)           # reference :73-83
        assert max_gen_len <= self.max_seq_len, (
            f"Generator can generate up to {self.max_seq_len} sequences, max_gen_len={max_gen_len}")
        self.max_gen_len = max_gen_len
        min_gen_len = min_gen_len or self.min_seq_len
        assert min_gen_len is not None, "A `min_gen_len` is required"
        assert min_gen_len > 0, f"min_gen_len must be greater than or equal to 1, min_gen_len={min_gen_len}"
        self.min_gen_len = min_gen_len
        self.temperature = temperature if temperature != 0.0 else opt.lcm_temperature

        self.seqs = seqs
        self.preds = torch.zeros((B, T - min_context_len, D), device=seqs.device, dtype=model.dtype)
        self.step_scores = torch.zeros((B, T), device=seqs.device)
        self.hypotheses: List[List[Hypothesis]] = [[] for _ in range(B)]
        if self.one_pass and self._can_score_in_one_pass():
            self._score_one_pass(text_lens, noise)
        else:
            self._score_step_by_step(text_lens, noise, disable_cache)
        return SequenceGeneratorOutput(self.hypotheses, counters=GenerationCounters())

    def _finish(self, d: int, text_len: int) -> None:
        """reference `finish_sequence`, :171-181"""
        self.hypotheses[d] = [Hypothesis(seq=self.preds[d, :text_len - self.min_context_len], score=None,
                                         step_scores=self.step_scores[d])]

    # ---- base LCM: one pass ----------------------------------------------------------------------------------
    def _can_score_in_one_pass(self) -> bool:
        return True

    def _score_one_pass(self, text_lens: List[int], noise: Optional[Tensor]) -> None:
        B, T, D = self.seqs.shape
        mcl = self.min_context_len
        if T <= mcl:
            return
        opt = self.options
        nz = None
        if noise is not None:
            # the draw of step j (which predicts position j from positions < j) belongs to input position j - 1
            nz = torch.zeros((B, T - 1, D), device=self.seqs.device)
            nz[:, mcl - 1:] = noise[:T - mcl].reshape(T - mcl, B, D).transpose(0, 1)
        out = self.model.predict_next_sentence(EmbeddingsBatch(self.seqs[:, :T - 1], None), sample=opt.sample_latent_variable,
                                               temperature=self.temperature, state_bag=None, noise=nz)
        self.preds.copy_(out.seqs[:, mcl - 1:].to(self.preds.dtype))
        for d, n in enumerate(text_lens):        # every document finishes with its own number of scored positions
            self._finish(d, n)

    # ---- the reference's loop, shared with the two-tower scorer -----------------------------------------------------
    def _score_step_by_step(self, text_lens: List[int], noise: Optional[Tensor], disable_cache: bool) -> None:
        B, T, _ = self.seqs.shape
        mcl = self.min_context_len
        device = self.seqs.device
        self._new_state(T, disable_cache)
        active = list(range(B))
        active_t = torch.arange(B, device=device)
        try:
            self.step_nr = mcl
            prefill_len = mcl - 1
            if prefill_len > 0:
                self._prefill(self.seqs[:, :prefill_len])
            else:
                logger.warning("Skipping prefill since only a context size of %d is provided in the prefix", mcl)
            for step in range(mcl, T):
                self.step_nr = step
                everyone = len(active) == B
                prev = self.seqs[:, step - 1:step] if everyone else self.seqs[active_t, step - 1:step]
                nz = None
                if noise is not None:
                    nz = noise[step - mcl].reshape(B, -1)
                    nz = nz if everyone else nz[active_t]
                new = self._predict(prev, nz, active_t, everyone)                # (n_act, D) fp32
                if everyone:
                    self.preds[:, step - mcl] = new.to(self.preds.dtype)
                else:
                    self.preds[active_t, step - mcl] = new.to(self.preds.dtype)
                keep = [r for r, d in enumerate(active) if text_lens[d] > step + 1]
                if len(keep) != len(active):
                    for r, d in enumerate(active):
                        if text_lens[d] <= step + 1:
                            self._finish(d, text_lens[d])
                    if not keep:
                        break
                    active = [active[r] for r in keep]
                    active_t = torch.tensor(active, device=device)
                    self.reorder_state(keep)
        finally:
            self._release()
