"""Sentence-by-sentence generation with the two-tower diffusion LCM on the HIP path.

Same constructor, call signature, option names and output type as the reference's
lcm/inference/two_tower_diffusion_lcm/generator.py:29-213, and the same stopping rules as its
`finalize_step` (:305-423).  What differs is where state lives: the prompt/output buffer stays
indexed by document for the whole call, finished documents leave an *active list* kept on the
host, and the KV caches are never copied -- `reorder_state` (:442-466) becomes an update of the
session's row -> cache-slot map.  With no EOS vector and no repetition threshold (the benchmark
setting) a generation step issues no device->host synchronisation at all.

Extension: `noise=` (keyword, (steps, B, D) or (steps, B, 1, D)) supplies the N(0, 1) draws,
consumed as noise[step - min_prompt_len, document]; without it each step draws
`torch.randn(n_active, 1, D)` on the device like the reference (seeded by `options.seed`).
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
    ddim_eta: float = 0.0                 # 0 = deterministic DDIM (the only implemented value)
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

    @torch.inference_mode()
    def __call__(self, batch_input: EmbeddingsBatch, max_gen_len: Optional[int] = None,
                 min_gen_len: Optional[int] = None, temperature: float = 0.0, disable_cache: bool = False,
                 **kwargs) -> SequenceGeneratorOutput:
        opt, model = self.options, self.model
        if opt.seed:
            torch.manual_seed(opt.seed)
        if disable_cache:
            # the reference asserts a context state bag in prefill() (generator.py:226-228)
            raise AssertionError("Expecting a context state bag to prefill")
        noise: Optional[Tensor] = kwargs.get("noise")

        seqs_in = batch_input.seqs
        B, P_max, D = seqs_in.shape
        device = seqs_in.device
        mask = batch_input.padding_mask
        if mask is None:
            P_min, prompt_lens, prompt_mask = P_max, None, None
        else:
            assert mask.seq_lens is not None, "Expecting a valid `self.prompt_seq_lens` Tensor, found `None`"
            prompt_lens = [int(v) for v in mask.seq_lens.tolist()]
            P_min = min(prompt_lens)
            prompt_mask = mask.materialize()

        max_gen_len = max_gen_len or self.max_seq_len
        assert max_gen_len <= self.max_seq_len, (
            f"Generator can generate up to {self.max_seq_len} sequences, max_gen_len={max_gen_len}")
        min_gen_len = min_gen_len or self.min_seq_len
        assert min_gen_len > 0, f"min_gen_len must be greater than or equal to 1, min_gen_len={min_gen_len}"
        self.max_gen_len, self.min_gen_len = max_gen_len, min_gen_len
        self.max_prompt_len, self.min_prompt_len = P_max, P_min
        self.temperature = temperature if temperature != 0.0 else opt.lcm_temperature
        total = P_max + max_gen_len

        self.state_bag = LCMIncrementalStateBag(total)
        self.context_state_bag = LCMIncrementalStateBag(total)
        self.seqs = torch.zeros((B, total, D), device=device, dtype=model.dtype)
        self.step_scores = torch.zeros((B, total), device=device)
        self.seqs[:, :P_max] = seqs_in
        self.hypotheses: List[List[Hypothesis]] = [[] for _ in range(B)]
        lengths = [-1] * B
        active = list(range(B))                       # documents still generating, ascending
        active_t = torch.arange(B, device=device)
        eos = self.eos_vec.to(device).float() if (self.eos_threshold is not None and self.eos_vec is not None) else None
        rep_thr = opt.stop_on_repetition_cosine_threshold

        try:
            # ---- prefill: every prompt position but the last one common to all documents
            prefill_len = P_min - 1
            if prefill_len > 0:
                self.context = model.encode(EmbeddingsBatch(self.seqs[:, :prefill_len], None),
                                            state_bag=self.context_state_bag, normalized=False)
                self.context_state_bag.increment_step_nr(prefill_len)
            else:
                logger.warning("Skipping prefill since only a context size of %d is provided in the prefix", P_min)
                self.context = None

            for step in range(P_min, total):
                self.step_nr = step
                n_act = len(active)
                everyone = n_act == B
                prev = self.seqs[:, step - 1:step] if everyone else self.seqs[active_t, step - 1:step]
                if noise is not None:
                    nz = noise[step - P_min].reshape(B, -1)
                    nz = nz if everyone else nz[active_t]
                else:
                    nz = None
                out, self.context = model.predict_next_sentence(
                    EmbeddingsBatch(prev, None), self.context, temperature=self.temperature,
                    state_bag=self.state_bag, context_state_bag=self.context_state_bag, noise=nz)
                self.state_bag.increment_step_nr()
                new = out.seqs[:, -1]                                        # (n_act, D) fp32

                # ---- stopping rules (host-side unless an EOS / repetition test needs the embeddings)
                in_prompt = [False] * n_act
                if step < P_max:
                    assert prompt_mask is not None
                    in_prompt = [prompt_lens[d] > step for d in active]
                    if any(in_prompt):
                        m = prompt_mask[:, step] if everyone else prompt_mask[active_t, step]
                        src = self.seqs[:, step] if everyone else self.seqs[active_t, step]
                        new = torch.where(m.unsqueeze(-1), src.float(), new)
                stop = [False] * n_act
                if eos is not None or (rep_thr is not None and step > 0):
                    hit = torch.zeros(n_act, dtype=torch.bool, device=device)
                    if eos is not None:
                        hit |= torch.nn.functional.cosine_similarity(eos, new).ge(self.eos_threshold)
                    if rep_thr is not None and step > 0:
                        hit |= torch.nn.functional.cosine_similarity(prev[:, 0].float(), new).ge(rep_thr)
                    stop = hit.tolist()                                      # the only sync of the step
                finished_now = []
                for r, d in enumerate(active):
                    plen = prompt_lens[d] if prompt_lens is not None else P_max
                    eos_stop = stop[r] and not in_prompt[r] and (step - plen) >= min_gen_len
                    if eos_stop or (step - plen + 1) >= max_gen_len:
                        lengths[d] = step + 1
                        finished_now.append((r, d, eos_stop))

                if everyone:
                    self.seqs[:, step] = new.to(self.seqs.dtype)
                else:
                    self.seqs[active_t, step] = new.to(self.seqs.dtype)

                if finished_now:
                    for _, d, eos_stop in finished_now:
                        n = lengths[d]
                        if opt.trim_hypotheses and eos_stop:
                            n -= int(not opt.include_eos_token)
                        self.hypotheses[d] = [Hypothesis(seq=self.seqs[d, :n], score=None, step_scores=self.step_scores[d])]
                    gone = {r for r, _, _ in finished_now}
                    keep = [r for r in range(n_act) if r not in gone]
                    if not keep:
                        break
                    active = [active[r] for r in keep]
                    active_t = torch.tensor(active, device=device)
                    self.reorder_state(keep)
        finally:
            model.release_session(self.context_state_bag.session)
            self.context_state_bag.session = None
        self.lengths = lengths
        return SequenceGeneratorOutput(self.hypotheses, counters=GenerationCounters())

    def reorder_state(self, new_order) -> None:
        """Drops finished rows from the device state: `new_order` are the surviving row indices."""
        self.state_bag.reorder(new_order)
        self.context_state_bag.reorder(new_order)
