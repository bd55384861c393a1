"""Sentence-by-sentence generation with an LCM on the HIP path.

Same constructor, call signature, option names and output type as the reference's
lcm/inference/lcm/generator.py:29-224, and the same stopping rules as its `finalize_step` (:288-406).  What
differs is where state lives: the prompt/output buffer stays indexed by document for the whole call, finished
documents leave an *active list* kept on the host, and the KV caches are never copied -- `reorder_state`
(:429-448) becomes an update of the session's row -> cache-slot map.  With no EOS vector and no repetition
threshold a generation step issues no device->host synchronisation at all.

`LCMGenerator` drives a base LCM (`BaseLCModel.predict_next_sentence`); the two-tower diffusion generator
overrides the three model hooks (`_prefill`, `_predict`, `_release`).

Extension: `noise=` (keyword, (steps, B, D) or (steps, B, 1, D)) supplies the N(0, 1) draws, consumed as
noise[step - min_prompt_len, document]; without it each step draws on the device like the reference (seeded by
`options.seed`).
"""

import logging
from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import Tensor

from lcm import _hip
from lcm.datasets.batch import EmbeddingsBatch
from lcm.inference.output import GenerationCounters, Hypothesis, SequenceGeneratorOutput
from lcm.nn.incremental_state import LCMIncrementalStateBag

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
        self.state_bag = None

    takes_option_overrides = True      # the base generator copies matching **kwargs onto its options (reference :181-183)

    # ---- model hooks (base LCM; the diffusion generator overrides them) ---------------------------------------
    def _prefill(self, prompt: Tensor) -> None:
        """prompt (B, prefill_len, D): fill the KV cache (reference :226-246; the prediction is discarded)."""
        if self.state_bag is None:
            return
        self.model.prefill(EmbeddingsBatch(prompt, None), self.state_bag)
        self.state_bag.increment_step_nr(prompt.shape[1])

    def _predict(self, prev: Tensor, noise: Optional[Tensor], active_t: Tensor, everyone: bool) -> Tensor:
        """prev (n_act, 1, D): the latest embedding of every active document -> next embedding (n_act, D) fp32
        (reference `_decode` / `_step`, :248-286)."""
        opt = self.options
        if self.state_bag is None:
            # no KV cache: the whole prefix goes through the model again (reference :270-275)
            full = self.seqs[:, :self.step_nr] if everyone else self.seqs[active_t, :self.step_nr]
            nz = None
            if noise is not None:
                nz = torch.zeros(full.shape, device=full.device)
                nz[:, -1] = noise
            out = self.model.predict_next_sentence(EmbeddingsBatch(full, None), sample=opt.sample_latent_variable,
                                                   temperature=self.temperature, state_bag=None, noise=nz)
            return out.seqs[:, -1]
        out = self.model.predict_next_sentence(EmbeddingsBatch(prev, None), sample=opt.sample_latent_variable,
                                               temperature=self.temperature, state_bag=self.state_bag,
                                               noise=None if noise is None else noise.unsqueeze(1))
        self.state_bag.increment_step_nr()
        return out.seqs[:, -1]

    def _release(self) -> None:
        if self.state_bag is not None:
            self.model.release_session(self.state_bag.session)
            self.state_bag.session = None

    def _new_state(self, total: int, disable_cache: bool) -> None:
        self.state_bag = None if disable_cache else LCMIncrementalStateBag(total)

    def reorder_state(self, new_order) -> None:
        """Drops finished rows from the device state: `new_order` are the surviving row indices."""
        if self.state_bag is not None:
            self.state_bag.reorder(new_order)

    # ---- the generation loop ---------------------------------------------------------------------------------
    @torch.inference_mode()
    def __call__(self, batch_input: EmbeddingsBatch, max_gen_len: Optional[int] = None,
                 min_gen_len: Optional[int] = None, temperature: float = 0.0, disable_cache: bool = False,
                 **kwargs) -> SequenceGeneratorOutput:
        opt, model = self.options, self.model
        if opt.seed:
            torch.manual_seed(opt.seed)
        noise: Optional[Tensor] = kwargs.pop("noise", None)
        # stochastic DDIM only: (generation steps, diffusion steps, documents, D) N(0, 1) draws, indexed like `noise`
        self._variance_noise: Optional[Tensor] = kwargs.pop("variance_noise", None)
        self._variance_step: Optional[Tensor] = None
        if self.takes_option_overrides:
            for k, v in kwargs.items():             # reference :181-183
                if hasattr(opt, k) and v:
                    setattr(opt, k, v)

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

        self._new_state(total, disable_cache)
        self.seqs = torch.zeros((B, total, D), device=device, dtype=model.dtype)
        self.step_scores = torch.zeros((B, total), device=device)
        self.seqs[:, :P_max] = seqs_in
        self.hypotheses: List[List[Hypothesis]] = [[] for _ in range(B)]
        lengths = [-1] * B
        active = list(range(B))                       # documents still generating, ascending
        active_t = torch.arange(B, device=device)
        eos = self.eos_vec.to(device).float().contiguous() if (self.eos_threshold is not None and self.eos_vec is not None) else None
        rep_thr = opt.stop_on_repetition_cosine_threshold
        seq_code = _hip.dtype_code(self.seqs.dtype)
        cos_buf = torch.zeros((B, 2), device=device, dtype=torch.float32)
        prompt_lens_dev = None if prompt_lens is None else torch.tensor(prompt_lens, device=device, dtype=torch.int32)
        active_i32 = active_t.to(torch.int32)

        try:
            # ---- prefill: every prompt position but the last one common to all documents
            self.step_nr = P_min
            prefill_len = P_min - 1
            if prefill_len > 0:
                self._prefill(self.seqs[:, :prefill_len])
            else:
                logger.warning("Skipping prefill since only a context size of %d is provided in the prefix", P_min)

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
                if self._variance_noise is not None:
                    vn = self._variance_noise[step - P_min]
                    self._variance_step = vn if everyone else vn[:, active_t]
                new = self._predict(prev, nz, active_t, everyone)            # (n_act, D) fp32

                # ---- finalize_step: one kernel keeps the prompt embedding of rows still inside their prompt, stores
                # every other row's new sentence into the sequence buffer and takes the two cosines of the stopping
                # rules; the host reads them (the only sync of the step) only when a rule needs them
                in_prompt = [False] * n_act
                if step < P_max and prompt_lens is not None:
                    in_prompt = [prompt_lens[d] > step for d in active]
                need_cos = eos is not None or (rep_thr is not None and step > 0)
                new32 = new.float().contiguous()
                _hip.check(_hip.lib.lcmhip_finalize_step(
                    _hip.ptr(new32), _hip.ptr(self.seqs), seq_code, None if everyone else _hip.ptr(active_i32),
                    _hip.ptr(prompt_lens_dev) if step < P_max else None, n_act, step, total, D, _hip.ptr(eos),
                    _hip.ptr(cos_buf), _hip.stream_ptr(device)))
                stop = [False] * n_act
                if need_cos:
                    cos = cos_buf[:n_act].tolist()
                    for r in range(n_act):
                        stop[r] = ((eos is not None and cos[r][0] >= self.eos_threshold)
                                   or (rep_thr is not None and step > 0 and cos[r][1] >= rep_thr))
                finished_now = []
                for r, d in enumerate(active):
                    plen = prompt_lens[d] if prompt_lens is not None else P_max
                    eos_stop = stop[r] and not in_prompt[r] and (step - plen) >= min_gen_len
                    if eos_stop or (step - plen + 1) >= max_gen_len:
                        lengths[d] = step + 1
                        finished_now.append((r, d, eos_stop))

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
                    active_i32 = active_t.to(torch.int32)
                    self.reorder_state(keep)
        finally:
            self._release()
        self.lengths = lengths
        return SequenceGeneratorOutput(self.hypotheses, counters=GenerationCounters())
