"""`EmbeddingsBatch` and `PaddingMask`: the containers the generator API exchanges.

Mirrors the fields of the reference's lcm/datasets/batch.py:43-70 and the part of fairseq2's
PaddingMask that callers of this path read (`seq_lens`, `materialize()`).
"""

import copy
from dataclasses import dataclass
from enum import Enum
from typing import Any, Dict, List, Optional, Tuple, Union

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

    def double_seqs(self) -> "EmbeddingsBatch":
        """Every position twice in a row (1, 2, 3 -> 1, 1, 2, 2, 3, 3), lengths and source lengths with it (reference
        batch.py:164-188); the diffusion timesteps and document lengths are carried over as they are."""
        mask = self.padding_mask
        return EmbeddingsBatch(
            self.seqs.repeat_interleave(2, dim=1),
            None if mask is None else PaddingMask(2 * mask.seq_lens, 2 * mask._batch_seq_len),
            self.diffusion_timesteps, self.document_lengths,
            None if self.source_lengths is None else self.source_lengths.repeat_interleave(2, dim=0))

    def flatten_to_sentences(self) -> Tensor:
        """(sum of lengths, D): the unpadded positions of all sequences, in batch order (reference batch.py:190-218)."""
        if self.padding_mask is None:
            return self.seqs.reshape(-1, self.seqs.size(-1))
        return self.seqs[self.padding_mask.materialize().to(self.seqs.device)]

    def _with_seqs(self, seqs: Tensor) -> "EmbeddingsBatch":
        return EmbeddingsBatch(seqs, self.padding_mask, self.diffusion_timesteps, self.document_lengths,
                               self.source_lengths)

    def normalize_seqs(self, normalizer) -> "EmbeddingsBatch":
        return self if normalizer is None else self._with_seqs(normalizer.normalize(self.seqs))

    def denormalize_seqs(self, normalizer) -> "EmbeddingsBatch":
        return self if normalizer is None else self._with_seqs(normalizer.denormalize(self.seqs))


class LCMStyle(Enum):
    """How an `LCMInput` becomes model input (reference batch.py:28-40)."""
    UNSUPERVISED = 0
    SUPERVISED = 1
    PACKED_UNSUPERVISED = 2


DOC_LENGTHS = "__doc_lengths"       # the column of a raw loader batch that holds the lengths of packed documents


def get_embeddings_sequence(src_seqs: List[Tensor], tgt_seqs: Optional[List[Tensor]] = None,
                            document_lengths: Optional[Tensor] = None, double_target: bool = False) -> EmbeddingsBatch:
    """Source (followed by target, optionally with every target position twice) per document, right-padded into one
    batch; `source_lengths` is filled in when there are targets (reference batch.py:381-425)."""
    rows = []
    for i, src in enumerate(src_seqs):
        if tgt_seqs is None or tgt_seqs[i] is None:
            rows.append(src)
            continue
        tgt = tgt_seqs[i].to(device=src.device, dtype=src.dtype)
        rows.append(torch.cat([src, tgt.repeat_interleave(2, dim=0) if double_target else tgt]))
    seqs, mask = pad_seqs(rows)
    return EmbeddingsBatch(
        seqs, mask,
        document_lengths=None if document_lengths is None else document_lengths.to(seqs.device),
        source_lengths=None if tgt_seqs is None else torch.tensor([s.size(0) for s in src_seqs], device=seqs.device))


@dataclass
class LCMInput:
    """Source / target documents of SONAR embeddings as the loaders hand them over (reference batch.py:221-378)."""
    source: List[Tensor]
    target: Optional[List[Tensor]] = None
    tokens: Any = None
    target_tokens: Any = None
    name: Optional[str] = None
    batch: Optional[Dict[str, Any]] = None

    def __post_init__(self) -> None:
        assert self.source is not None
        assert self.target is None or len(self.target) == len(self.source), (
            f"all elements in LCMInput should be of the same length, got {len(self.target)} and {len(self.source)}")

    def __len__(self) -> int:
        return len(self.source)

    def __getitem__(self, i: int) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        return self.source[i] if self.target is None else (self.source[i], self.target[i])

    def prepare_input(self, style: LCMStyle = LCMStyle.UNSUPERVISED) -> EmbeddingsBatch:
        if style == LCMStyle.UNSUPERVISED:
            return get_embeddings_sequence(self.source)
        if style == LCMStyle.PACKED_UNSUPERVISED:
            lengths = None if self.batch is None else self.batch.get(DOC_LENGTHS)
            if lengths is not None:
                assert len(lengths) == 1, "Expecting batch size of 1"       # packed documents come one row at a time
                lengths = lengths[0].type(torch.int64)
            return get_embeddings_sequence(self.source, document_lengths=lengths)
        if style == LCMStyle.SUPERVISED:
            assert self.target is not None, "Missing target embeddings for a supervised batch"
            return get_embeddings_sequence(self.source, self.target)
        raise ValueError(f"Unsupported style={style} - could not prepare input")

    def prepare_target_mask(self, embeddings: EmbeddingsBatch, style: LCMStyle,
                            min_context_size: Optional[int] = None) -> Tensor:
        """(B, S) bool: the positions a scorer or criterion judges.  Unsupervised: everything from `min_context_size`
        on (per packed document when the batch carries document lengths); supervised: everything after the source;
        padding is never a target."""
        n, width, _ = embeddings.seqs.shape
        device = embeddings.seqs.device
        if style == LCMStyle.SUPERVISED:
            src = torch.tensor([s.size(0) for s in self.source], device=device)
            mask = torch.arange(width, device=device).unsqueeze(0) >= src.unsqueeze(1)
        elif style == LCMStyle.PACKED_UNSUPERVISED and embeddings.document_lengths is not None:
            start = torch.cat([torch.arange(int(m), device=device) for m in embeddings.document_lengths])
            mask = (start >= (min_context_size or 0)).unsqueeze(0)
        elif style in (LCMStyle.UNSUPERVISED, LCMStyle.PACKED_UNSUPERVISED):
            mask = torch.ones(n, width, dtype=torch.bool, device=device)
            if min_context_size is not None:
                mask[:, :min_context_size] = False
        else:
            raise ValueError(f"Unsupported style={style}")
        if embeddings.padding_mask is not None:
            mask = mask & embeddings.padding_mask.materialize().to(device)
        return mask
