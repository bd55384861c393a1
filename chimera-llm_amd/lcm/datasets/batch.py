"""`EmbeddingsBatch` and `PaddingMask`: the containers the generator API exchanges.

Mirrors the fields of the reference's lcm/datasets/batch.py:43-70 and the part of fairseq2's
PaddingMask that callers of this path read (`seq_lens`, `materialize()`).
"""

from dataclasses import dataclass
from typing import List, Optional

import torch
from torch import Tensor


class PaddingMask:
    """Sequence lengths of a right-padded batch.  `materialize()` is True on real positions."""

    def __init__(self, seq_lens: Tensor, batch_seq_len: int) -> None:
        self._seq_lens = seq_lens
        self._batch_seq_len = int(batch_seq_len)

    @property
    def seq_lens(self) -> Tensor:
        return self._seq_lens

    def materialize(self) -> Tensor:
        steps = torch.arange(self._batch_seq_len, device=self._seq_lens.device)
        return steps.unsqueeze(0) < self._seq_lens.unsqueeze(1)

    def to(self, device) -> "PaddingMask":
        return PaddingMask(self._seq_lens.to(device), self._batch_seq_len)


def pad_seqs(seqs: List[Tensor], pad_value: float = 0.0):
    """Right-pads (len_i, D) tensors to (B, max_len, D); the mask is None when nothing is padded."""
    lens = torch.tensor([s.shape[0] for s in seqs], device=seqs[0].device)
    width = int(lens.max())
    out = seqs[0].new_full((len(seqs), width) + tuple(seqs[0].shape[1:]), pad_value)
    for i, s in enumerate(seqs):
        out[i, : s.shape[0]] = s
    if bool((lens == width).all()):
        return out, None
    return out, PaddingMask(lens, width)


@dataclass
class EmbeddingsBatch:
    seqs: Tensor                                  # (B, S, D)
    padding_mask: Optional[PaddingMask] = None
    diffusion_timesteps: Optional[Tensor] = None  # (B, S)
    document_lengths: Optional[Tensor] = None
    source_lengths: Optional[Tensor] = None

    def __len__(self) -> int:
        return self.seqs.size(0)

    @property
    def batch_size(self) -> int:
        return self.seqs.size(0)

    @property
    def shape(self) -> torch.Size:
        return self.seqs.shape

    @property
    def device(self):
        return self.seqs.device

    def unbatch(self) -> List[Tensor]:
        if self.padding_mask is None:
            return list(self.seqs)
        return [row[: int(n)] for row, n in zip(self.seqs, self.padding_mask.seq_lens)]

    def _with_seqs(self, seqs: Tensor) -> "EmbeddingsBatch":
        return EmbeddingsBatch(seqs, self.padding_mask, self.diffusion_timesteps, self.document_lengths,
                               self.source_lengths)

    def normalize_seqs(self, normalizer) -> "EmbeddingsBatch":
        return self if normalizer is None else self._with_seqs(normalizer.normalize(self.seqs))

    def denormalize_seqs(self, normalizer) -> "EmbeddingsBatch":
        return self if normalizer is None else self._with_seqs(normalizer.denormalize(self.seqs))
