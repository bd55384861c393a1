"""The embedding-level half of the reference's evaluation predictor (lcm/evaluation/predictors/lcm.py:205-262):
a list of already-encoded prompts in, one generated continuation per prompt out.  The text half of that class (SONAR
encoding of the prompts, SONAR decoding of the hypotheses) needs the SONAR models and is not part of this package."""

from typing import List, Optional, Sequence

import torch
from torch import Tensor

from lcm.datasets.parquet_prompts import get_embeddings_sequence


@torch.inference_mode()
def generate_continuations(generator, prompts: Sequence[Tensor], *, max_gen_len: Optional[int] = None,
                           min_gen_len: Optional[int] = None, max_gen_len_ratio: Optional[float] = None,
                           max_prompt_len: Optional[int] = None, generator_batch_size: Optional[int] = None,
                           temperature: float = 0.0, disable_cache: bool = False, **generator_kwargs) -> List[Tensor]:
    """`prompts`: one (sentences, D) tensor per document, on the generator's device.  They are cut to `max_prompt_len`
    sentences, taken `generator_batch_size` at a time (all at once when it is None), padded into an `EmbeddingsBatch`
    and handed to `generator`; with `max_gen_len_ratio` a batch generates at most that multiple of its longest prompt
    (at least one sentence).  Returns, in the order of `prompts`, the generated sentences of every document
    (generated, D): the best hypothesis without its prompt."""
    seqs = [p[:max_prompt_len] for p in prompts] if max_prompt_len else list(prompts)
    assert all(isinstance(p, Tensor) and p.dim() == 2 and p.shape[0] >= 1 for p in seqs), \
        "Expect sonarized prompts in the form of a list of (sentences, D) tensors"
    step = generator_batch_size or max(len(seqs), 1)
    out: List[Tensor] = []
    for i in range(0, len(seqs), step):
        chunk = seqs[i:i + step]
        batch = get_embeddings_sequence(src_seqs=chunk)
        gen_len = max_gen_len
        if max_gen_len_ratio:
            gen_len = max(1, int(max(p.shape[0] for p in chunk) * max_gen_len_ratio))
        result = generator(batch, max_gen_len=gen_len, min_gen_len=min_gen_len, temperature=temperature,
                           disable_cache=disable_cache, **generator_kwargs)
        for prompt, hyps in zip(chunk, result.hypotheses):
            out.append(hyps[0].seq[prompt.shape[0]:])
    return out
