"""Generator host loop restated (test infrastructure).

Follows lcm/inference/two_tower_diffusion_lcm/generator.py:29-73 (options), :98-213 (__call__),
:222-259 (prefill), :261-303 (_decode/_step), :305-423 (finalize_step), :425-466
(finish_sequence / reorder_state) and the option defaults of lcm/inference/lcm/generator.py:29-58.
The in-model `torch.randn` is replaced by host-supplied noise indexed by [generation step,
original document index] so that results do not depend on batch compaction.
"""

from dataclasses import dataclass
from typing import List, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from .model import OracleBaseLCM, KVCache, OracleTwoTower


@dataclass
class GenOptions:
    max_seq_len: int = 200
    min_seq_len: int = 1
    eos_threshold: Optional[float] = 0.9
    stop_on_repetition_cosine_threshold: Optional[float] = None
    sample_latent_variable: bool = True
    include_eos_token: bool = False
    trim_hypotheses: bool = False
    seed: Optional[int] = None
    lcm_temperature: float = 1.0
    guidance_scale: float = 1.0
    guidance_rescale: float = 0.0
    ddim_eta: float = 0.0
    epsilon_scaling: Optional[float] = None
    initial_noise_scale: float = 1.0
    inference_timesteps: int = 100
    clip_noise: int = 100
    thresholding: bool = False
    dynamic_thresholding_ratio: float = 0.995
    sample_max_value: float = 6.0


@dataclass
class Hyp:
    seq: Tensor
    step_scores: Tensor


class OracleGenerator:
    def __init__(self, model: OracleTwoTower, options: Optional[GenOptions] = None,
                 eos_vec: Optional[Tensor] = None):
        self.model = model
        self.options = options or GenOptions()
        self.eos_vec = eos_vec
        assert self.options.min_seq_len >= 1
        model.prep_for_denoising(self.options)

    @torch.inference_mode()
    def __call__(self, seqs_in: Tensor, prompt_lens: Optional[Tensor] = None, max_gen_len: Optional[int] = None,
                 min_gen_len: Optional[int] = None, noise: Optional[Tensor] = None,
                 trace: Optional[list] = None, temperature: float = 0.0,
                 variance_noise: Optional[Tensor] = None) -> List[List[Hyp]]:
        """seqs_in (B,P,D) padded prompts; prompt_lens (B,) or None; noise (steps,B,D) consumed as
        noise[step_nr - min_prompt_len, original document]; variance_noise (steps, diffusion steps, B, D) likewise
        (stochastic DDIM only)."""
        opt, model = self.options, self.model
        self.temperature = temperature if temperature != 0.0 else opt.lcm_temperature   # lcm/generator.py:174-179
        B, P_max, D = seqs_in.shape
        if prompt_lens is None:
            P_min, pmask = P_max, None
        else:
            P_min = int(prompt_lens.min())
            pmask = torch.arange(P_max)[None, :] < prompt_lens[:, None]          # True = real token
        max_gen_len = max_gen_len or opt.max_seq_len
        assert max_gen_len <= opt.max_seq_len
        min_gen_len = min_gen_len or opt.min_seq_len
        assert min_gen_len > 0
        total = P_max + max_gen_len

        cache = KVCache(total)
        seqs = torch.zeros(B, total, D)
        step_scores = torch.zeros(B, total)
        lengths = torch.full((B,), -1, dtype=torch.int)
        sample_idx = torch.arange(B)
        hyps: List[List[Hyp]] = [[] for _ in range(B)]
        seqs[:, :P_max] = seqs_in

        # prefill (generator.py:222-259): everything but the last common prompt position
        prefill_len = P_min - 1
        base = isinstance(model, OracleBaseLCM)     # lcm/inference/lcm/generator.py:226-246 (same prefill rule)
        context = None if base else torch.empty(B, 0, model.model_dim)
        if prefill_len > 0:
            if base:
                model.predict(seqs[:, :prefill_len], cache)                   # output discarded, cache filled
            else:
                context = model.encode(model.normalize(seqs[:, :prefill_len]), cache)
            cache.increment_step_nr(prefill_len)

        for step_nr in range(P_min, total):
            n_act = seqs.shape[0]
            if noise is not None:
                nz = noise[step_nr - P_min].index_select(0, sample_idx).unsqueeze(1)
            else:
                nz = torch.randn(n_act, 1, D)
            if base:
                # lcm/inference/lcm/generator.py:248-286: one position through the cached decoder
                out = model.predict(seqs[:, step_nr - 1:step_nr], cache, nz, sample=opt.sample_latent_variable,
                                    temperature=self.temperature)
                cache.increment_step_nr()
            else:
                vn = None
                if variance_noise is not None:
                    vn = variance_noise[step_nr - P_min].index_select(1, sample_idx).unsqueeze(2)
                out, context = model.predict_next_sentence(seqs[:, step_nr - 1:step_nr], context, cache, nz,
                                                           trace=trace, variance_noise=vn)
            last = out[:, -1]

            # ---- finalize_step (generator.py:305-423)
            already = lengths > -1
            finish = torch.zeros_like(already)
            keep_going = None
            if step_nr < P_max:
                m = pmask[:, step_nr]
                last[m] = seqs[m, step_nr]
                keep_going = m
            if opt.eos_threshold is not None and self.eos_vec is not None:
                finish = finish | F.cosine_similarity(self.eos_vec, last).ge(opt.eos_threshold)
            if opt.stop_on_repetition_cosine_threshold is not None and step_nr > 0:
                finish = finish | F.cosine_similarity(seqs[:, step_nr - 1], last).ge(
                    opt.stop_on_repetition_cosine_threshold)
            if keep_going is not None:
                finish = finish & ~keep_going
            if prompt_lens is not None:
                finish = finish & (step_nr - prompt_lens).ge(min_gen_len)
                too_long = (step_nr - prompt_lens + 1).ge(max_gen_len)
            else:
                finish = finish & ((step_nr - P_max) >= min_gen_len)
                too_long = torch.full_like(finish, (step_nr - P_max + 1) >= max_gen_len)
            stopped_on_eos = finish
            finish = (finish | too_long) & ~already
            lengths[finish] = step_nr + 1
            seqs[:, step_nr] = last
            done = lengths.ne(-1)
            for idx in done.nonzero().flatten().tolist():
                n = int(lengths[idx])
                if opt.trim_hypotheses and n > -1 and bool(stopped_on_eos[idx]):
                    n = n - int(not opt.include_eos_token)
                hyps[int(sample_idx[idx])] = [Hyp(seq=seqs[idx, :n].clone(), step_scores=step_scores[idx].clone())]
            active = (~done).nonzero().flatten()
            if len(active) == 0:
                break
            # ---- reorder_state (generator.py:442-466)
            cache.reorder(active)
            if context is not None:
                context = context.index_select(0, active)
            seqs = seqs.index_select(0, active)
            sample_idx = sample_idx.index_select(0, active)
            step_scores = step_scores.index_select(0, active)
            lengths = lengths.index_select(0, active)
            if pmask is not None:
                pmask = pmask.index_select(0, active)
                prompt_lens = prompt_lens.index_select(0, active)
        return hyps


class OracleScorer(OracleGenerator):
    """Teacher-forced scoring, lcm/inference/two_tower_diffusion_lcm/scorer.py:27-314 and, for a base LCM,
    lcm/inference/lcm/scorer.py:19-198 (the same loop on the base generator's prefill / decode, lcm/generator.py:226-286):
    at every step the model sees the gold previous sentence, the prediction is recorded (never fed back), and a
    document leaves the batch after its last sentence."""

    @torch.inference_mode()
    def __call__(self, seqs_in: Tensor, text_lens: Optional[Tensor] = None, max_gen_len: Optional[int] = None,
                 min_gen_len: Optional[int] = None, min_context_len: int = 1, noise: Optional[Tensor] = None,
                 temperature: float = 0.0) -> List[List[Hyp]]:
        opt, model = self.options, self.model
        self.temperature = temperature if temperature != 0.0 else opt.lcm_temperature
        B, T, D = seqs_in.shape
        text_lens = torch.full((B,), T, dtype=torch.long) if text_lens is None else text_lens.clone()
        max_gen_len = min(max_gen_len or opt.max_seq_len, T - min_context_len)       # scorer.py:84-94
        assert max_gen_len <= opt.max_seq_len
        min_gen_len = min_gen_len or opt.min_seq_len
        assert min_gen_len > 0
        cache = KVCache(T)
        seqs = seqs_in.clone()
        preds = torch.zeros(B, T - min_context_len, D)
        step_scores = torch.zeros(B, T)
        sample_idx = torch.arange(B)
        hyps: List[List[Hyp]] = [[] for _ in range(B)]
        prefill_len = min_context_len - 1                                             # scorer.py:163-199
        base = isinstance(model, OracleBaseLCM)
        context = None if base else torch.empty(B, 0, model.model_dim)
        if prefill_len > 0:
            if base:
                model.predict(seqs[:, :prefill_len], cache)                           # lcm/generator.py:226-246
            else:
                context = model.encode(model.normalize(seqs[:, :prefill_len]), cache)
            cache.increment_step_nr(prefill_len)
        for step_nr in range(min_context_len, T):
            n_act = seqs.shape[0]
            nz = (noise[step_nr - min_context_len].index_select(0, sample_idx).unsqueeze(1) if noise is not None
                  else torch.randn(n_act, 1, D))
            if base:
                out = model.predict(seqs[:, step_nr - 1:step_nr], cache, nz, sample=opt.sample_latent_variable,
                                    temperature=self.temperature)                     # lcm/generator.py:248-286
                cache.increment_step_nr()
            else:
                out, context = model.predict_next_sentence(seqs[:, step_nr - 1:step_nr], context, cache, nz)
            # ---- finalize_step (scorer.py:243-280; lcm/scorer.py:135-169)
            keep_going = text_lens.gt(step_nr + 1)
            preds[:, step_nr - min_context_len] = out[:, -1]
            for idx in (~keep_going).nonzero().flatten().tolist():
                n = int(text_lens[idx])
                hyps[int(sample_idx[idx])] = [Hyp(seq=preds[idx, :n - min_context_len].clone(), step_scores=step_scores[idx].clone())]
            active = keep_going.nonzero().flatten()
            if len(active) == 0:
                break
            cache.reorder(active)                                                     # scorer.py:292-314; lcm/scorer.py:183-198
            if not base:
                context = context.index_select(0, active)
            seqs, preds = seqs.index_select(0, active), preds.index_select(0, active)
            sample_idx, step_scores = sample_idx.index_select(0, active), step_scores.index_select(0, active)
            text_lens = text_lens.index_select(0, active)
        return hyps
