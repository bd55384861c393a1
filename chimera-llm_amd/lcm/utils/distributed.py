"""Document-parallel generation across the GPUs of a node.

Documents never interact (attention, stopping and caches are per document), so ranks generate
disjoint blocks of documents with no collective inside the loop; the only exchange is ONE
all-gather of the finished embeddings together with their lengths.  The reference shards evaluation the same way, by
dataset shard per rank (lcm/evaluation/utils/data_utils.py:376-405).
"""

import math
from typing import Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_bounds(n_docs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `n_docs` documents owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_docs, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_shard(n_docs: int, world_size: int) -> int:
    """Largest block `shard_bounds` hands to any rank."""
    return -(-n_docs // world_size)


def _round16(n: int) -> int:
    return (n + 15) // 16 * 16


def all_gather_generations(seqs: Tensor, lengths: Tensor, group=None, width: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """seqs (B_loc, L, D) padded generations and lengths (B_loc,) of this rank -> the same for all ranks,
    concatenated in rank order.  Ranks may hold different B_loc <= `width`, the largest shard of any rank
    (`max_shard(n_docs, world)` for shards cut by `shard_bounds`; every rank must pass the same value).

    ONE collective: each rank packs [B_loc | lengths | sequences] into one byte buffer of the same size on every
    rank (shards shorter than `width` are padded) and a single all-gather moves it.  Without `width` the ranks first
    agree on it with one more (8-byte) all-reduce."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return seqs, lengths
    return _gather_packed(seqs, lengths, group, width)


def _gather_packed(seqs: Tensor, lengths: Tensor, group=None, width: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """The exchange itself (any world size, one rank included): pack, one all-gather, unpack."""
    world = dist.get_world_size(group)
    n_loc = seqs.shape[0]
    if width is None:
        w = torch.tensor([n_loc], device=seqs.device, dtype=torch.int64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX, group=group)
        width = int(w.item())
    assert n_loc <= width and lengths.shape[0] == n_loc, (n_loc, width, tuple(lengths.shape))
    row = math.prod(seqs.shape[1:]) * seqs.element_size()
    len_bytes = _round16(width * 4)
    nbytes = 16 + len_bytes + _round16(width * row)
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=seqs.device)
    buf[:4].copy_(torch.tensor([n_loc], dtype=torch.int32).view(torch.uint8))
    buf[16:16 + 4 * n_loc].copy_(lengths.to(torch.int32).contiguous().view(torch.uint8))
    buf[16 + len_bytes:16 + len_bytes + n_loc * row].copy_(seqs.contiguous().view(torch.uint8).reshape(-1))
    out = torch.empty(world * nbytes, dtype=torch.uint8, device=seqs.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.view(world, nbytes)
    counts = out[:, :4].contiguous().view(torch.int32).reshape(world).tolist()
    all_seqs, all_lens = [], []
    for r, c in enumerate(counts):
        all_lens.append(out[r, 16:16 + 4 * c].contiguous().view(torch.int32).to(lengths.dtype))
        all_seqs.append(out[r, 16 + len_bytes:16 + len_bytes + c * row].contiguous().view(seqs.dtype)
                        .reshape((c,) + tuple(seqs.shape[1:])))
    return torch.cat(all_seqs), torch.cat(all_lens)
