"""Document-parallel generation across the GPUs of a node.

Documents never interact (attention, stopping and caches are per document), so ranks generate
disjoint blocks of documents with no collective inside the loop; the only exchange is one
all-gather of the finished embeddings.  The reference shards evaluation the same way, by
dataset shard per rank (lcm/evaluation/utils/data_utils.py:376-405).
"""

from typing import List, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_bounds(n_docs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `n_docs` documents owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_docs, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def all_gather_generations(seqs: Tensor, lengths: Tensor, group=None) -> Tuple[Tensor, Tensor]:
    """seqs (B_loc, L, D) padded generations and lengths (B_loc,) of this rank -> the same for all
    ranks, concatenated in rank order.  Ranks may hold different B_loc."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return seqs, lengths
    world = dist.get_world_size(group)
    n_loc = torch.tensor([seqs.shape[0]], device=seqs.device, dtype=torch.int64)
    counts = [torch.zeros_like(n_loc) for _ in range(world)]
    dist.all_gather(counts, n_loc, group=group)
    counts = [int(c.item()) for c in counts]
    if len(set(counts)) == 1:
        out = seqs.new_empty((world * counts[0],) + tuple(seqs.shape[1:]))
        dist.all_gather_into_tensor(out, seqs.contiguous(), group=group)
        lens = lengths.new_empty(world * counts[0])
        dist.all_gather_into_tensor(lens, lengths.contiguous(), group=group)
        return out, lens
    width = max(counts)
    pad = seqs.new_zeros((width,) + tuple(seqs.shape[1:]))
    pad[: seqs.shape[0]] = seqs
    lpad = lengths.new_zeros(width)
    lpad[: lengths.shape[0]] = lengths
    bufs = [torch.empty_like(pad) for _ in range(world)]
    lbufs = [torch.empty_like(lpad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    dist.all_gather(lbufs, lpad, group=group)
    return (torch.cat([b[:c] for b, c in zip(bufs, counts)]), torch.cat([b[:c] for b, c in zip(lbufs, counts)]))
