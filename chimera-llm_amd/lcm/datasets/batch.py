"""`EmbeddingsBatch` and `PaddingMask`: the containers the generator API exchanges.

Mirrors the fields of the reference's lcm/datasets/batch.py:43-70 and the part of fairseq2's
PaddingMask that callers of this path read (`seq_lens`, `materialize()`).
"""

import copy
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

    def __post_init__(self) -> None:
        # reference batch.py:73-82: packed documents must fill the sequence (or half of it, for the doubled sequences
        # of diffusion training)
        if self.document_lengths is not None:
            total, width = int(self.document_lengths.sum()), self.seqs.size(1)
            assert total == width or 2 * total == width, (
                f"document_lengths {self.document_lengths.tolist()} sum to {total}, but the sequences are "
                f"{tuple(self.seqs.shape)} (neither their length nor half of it)")

    def __len__(self) -> int:
        return self.seqs.size(0)

    def __getitem__(self, i: int):
        # reference batch.py:104-107 refuses item access too
        raise NotImplementedError("Access to each item in EmbeddingsBatch not allowed yet")

    def clone(self) -> "EmbeddingsBatch":
        """A deep copy (reference batch.py:101-102)."""
        return copy.deepcopy(self)

    def get_last_element(self) -> Tensor:
        """(B, D): the last unpadded embedding of every sequence (reference batch.py:117-124)."""
        if self.padding_mask is None:
            return self.seqs[:, -1]
        last = self.padding_mask.seq_lens.to(self.seqs.device).long() - 1
        return self.seqs[torch.arange(self.seqs.size(0), device=self.seqs.device), last]

    def set_last_element(self, element: Tensor) -> None:
        """Overwrites the last unpadded embedding of every sequence in place (reference batch.py:126-132)."""
        element = element.to(self.seqs.device)
        if self.padding_mask is None:
            self.seqs[:, -1] = element
            return
        last = self.padding_mask.seq_lens.to(self.seqs.device).long() - 1
        self.seqs[torch.arange(self.seqs.size(0), device=self.seqs.device), last] = element.to(self.seqs.dtype)

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
