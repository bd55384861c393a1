"""Prompt ingestion: SONAR-embedded documents stored as parquet -> padded `EmbeddingsBatch`es on the GPU.

The data format is the reference's (lcm/datacards/datacards.yaml:1-20): one row per document, an embedding column
such as `text_sentences_sonar_emb` or `prompt_sentences_sonar_emb` holding a list of 1024-d float vectors (one
per sentence), optional text columns, hive-style `split=` partitions.  `get_embeddings_sequence` follows
lcm/datasets/batch.py:381-425 (source [+ target] concatenation, padding, `source_lengths`), the rank sharding
follows lcm/evaluation/utils/data_utils.py:376-405 (contiguous by rank, no synchronisation between workers).
Host-side code: pyarrow decodes on the CPU, one host->device copy per batch.
"""

from typing import Iterator, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from lcm.datasets.batch import EmbeddingsBatch, pad_seqs
from lcm.utils.distributed import shard_bounds


def _column_to_tensors(column, dim: int) -> List[Tensor]:
    """A pyarrow ChunkedArray of list<fixed_size_list<float>[dim]> or list<list<float>> -> (len_i, dim) fp32."""
    import numpy as np
    out: List[Tensor] = []
    for chunk in column.chunks:
        lens = chunk.value_lengths().fill_null(0).to_numpy(zero_copy_only=False)   # sentences per document
        vectors = chunk.flatten()                 # the chunk's sentence vectors (honours the chunk's own offsets)
        vals = np.asarray(vectors.flatten().to_numpy(zero_copy_only=False), dtype=np.float32)
        if vals.size != int(lens.sum()) * dim:
            raise ValueError(f"embedding column does not hold {dim}-d vectors")
        mat = torch.from_numpy(vals.reshape(-1, dim).copy())
        out.extend(mat.split([int(n) for n in lens]))
    return out


def read_sonar_documents(path: str, column: str, *, split: Optional[str] = None, dim: int = 1024,
                         text_column: Optional[str] = None, max_documents: Optional[int] = None,
                         rank: int = 0, world_size: int = 1) -> Tuple[List[Tensor], Optional[List[List[str]]]]:
    """Reads the documents owned by `rank` (contiguous block) from a parquet file or dataset directory."""
    import pyarrow.dataset as ds
    dataset = ds.dataset(path, format="parquet", partitioning="hive")
    flt = None
    if split is not None:
        flt = ds.field("split") == split
    cols = [column] + ([text_column] if text_column else [])
    table = dataset.to_table(columns=cols, filter=flt)
    n = table.num_rows if max_documents is None else min(table.num_rows, max_documents)
    lo, hi = shard_bounds(n, rank, world_size)
    table = table.slice(lo, hi - lo)
    docs = _column_to_tensors(table.column(column), dim)
    texts = table.column(text_column).to_pylist() if text_column else None
    return docs, texts


def get_embeddings_sequence(src_seqs: Sequence[Tensor], tgt_seqs: Optional[Sequence[Tensor]] = None,
                            document_lengths: Optional[Tensor] = None, double_target: bool = False) -> EmbeddingsBatch:
    """Ragged documents -> one padded batch (reference: lcm/datasets/batch.py:381-425)."""
    seqs_lst: List[Tensor] = []
    for i, src in enumerate(src_seqs):
        parts = [src]
        if tgt_seqs is not None and tgt_seqs[i] is not None:
            tgt = tgt_seqs[i].to(src.device).type(src.dtype)
            parts.append(torch.repeat_interleave(tgt, 2, dim=0) if double_target else tgt)
        seqs_lst.append(torch.concat(parts))
    seqs, padding_mask = pad_seqs(seqs_lst)
    source_lengths = None
    if tgt_seqs is not None:
        source_lengths = torch.tensor([s.size(0) for s in src_seqs], device=seqs.device)
    if document_lengths is not None:
        document_lengths = document_lengths.to(seqs.device)
    return EmbeddingsBatch(seqs, padding_mask=padding_mask, document_lengths=document_lengths,
                           source_lengths=source_lengths)


def iterate_prompt_batches(path: str, column: str, batch_size: int, *, device="cuda", dtype=torch.float32,
                           split: Optional[str] = None, max_prompt_len: Optional[int] = None,
                           max_documents: Optional[int] = None, rank: int = 0, world_size: int = 1,
                           prefix_embeddings: Optional[Tensor] = None,
                           suffix_embeddings: Optional[Tensor] = None) -> Iterator[EmbeddingsBatch]:
    """Padded prompt batches of this rank's documents, ready for `generator(batch)`.  Prompts longer than
    `max_prompt_len` keep their first `max_prompt_len` sentences.  `prefix_embeddings` / `suffix_embeddings` (k, D):
    sentence embeddings put before / after every prompt, the way the reference's loader wraps a source in its
    `source_prefix_text` / `source_suffix_text` (lcm/datasets/dataloading.py:597-613); the reference encodes those
    texts with SONAR itself, here the caller brings them already encoded.  The length cut applies to the document, not
    to the affixes."""
    docs, _ = read_sonar_documents(path, column, split=split, max_documents=max_documents, rank=rank, world_size=world_size)
    if max_prompt_len is not None:
        docs = [d[:max_prompt_len] for d in docs]
    docs = [d for d in docs if d.shape[0] > 0]
    for name, t in (("prefix_embeddings", prefix_embeddings), ("suffix_embeddings", suffix_embeddings)):
        if t is not None and (t.dim() != 2 or (docs and t.shape[1] != docs[0].shape[1])):
            raise ValueError(f"{name} must be (sentences, {docs[0].shape[1] if docs else 'D'}), got {tuple(t.shape)}")

    def wrap(d: Tensor) -> Tensor:
        parts = [t.to(d.dtype) for t in (prefix_embeddings, d, suffix_embeddings) if t is not None]
        return d if len(parts) == 1 else torch.cat(parts)

    for i in range(0, len(docs), batch_size):
        chunk = [wrap(d).to(device=device, dtype=dtype, non_blocking=True) for d in docs[i:i + batch_size]]
        yield get_embeddings_sequence(chunk)
