"""World-size-2 gloo test of the document sharding + output all-gather used by bench.py --gpus N."""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, uneven, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from lcm.utils.distributed import all_gather_generations, max_shard, shard_bounds
    n_docs = 5 if uneven else 4
    lo, hi = shard_bounds(n_docs, rank, world)
    # "generation" of document g: rows filled with g, length g + 2
    seqs = torch.stack([torch.full((7, 8), float(g)) for g in range(lo, hi)]).to(torch.bfloat16)
    lens = torch.tensor([g + 2 for g in range(lo, hi)], dtype=torch.int32)
    # count the collectives the exchange issues: exactly one all-gather when the shard width is known
    calls = []
    for name in ("all_gather_into_tensor", "all_gather", "all_reduce", "broadcast", "all_to_all", "gather"):
        real = getattr(dist, name)
        setattr(dist, name, (lambda real, name: lambda *a, **k: (calls.append(name), real(*a, **k))[1])(real, name))
    all_seqs, all_lens = all_gather_generations(seqs, lens, width=max_shard(n_docs, world))
    assert calls == ["all_gather_into_tensor"], calls
    # without the width the ranks agree on it first: one more tiny all-reduce, same result
    again, lens_again = all_gather_generations(seqs, lens)
    assert calls == ["all_gather_into_tensor", "all_reduce", "all_gather_into_tensor"], calls
    assert torch.equal(again, all_seqs) and torch.equal(lens_again, all_lens)
    assert all_seqs.dtype == torch.bfloat16 and all_lens.dtype == torch.int32
    torch.save((all_seqs.float(), all_lens), os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _run(uneven, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, uneven, str(tmp_path)), nprocs=2, join=True)
    n_docs = 5 if uneven else 4
    for r in range(2):
        seqs, lens = torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=True)
        assert seqs.shape == (n_docs, 7, 8)
        assert lens.tolist() == [g + 2 for g in range(n_docs)]
        assert seqs[:, 0, 0].tolist() == [float(g) for g in range(n_docs)]      # rank order == document order


def test_all_gather_even_shards(tmp_path):
    _run(False, tmp_path)


def test_all_gather_uneven_shards(tmp_path):
    _run(True, tmp_path)


def test_shard_bounds_cover_everything():
    from lcm.utils.distributed import shard_bounds
    for n in (1, 7, 64, 512):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
