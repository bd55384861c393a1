"""Prompt ingestion from the reference's parquet layout (CPU; needs no GPU)."""

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest
import torch

from lcm.datasets.parquet_prompts import get_embeddings_sequence, iterate_prompt_batches, read_sonar_documents


def _write(path, lens, fixed, seed=0, split=None):
    rng = np.random.default_rng(seed)
    docs = [rng.standard_normal((n, 1024)).astype(np.float32) for n in lens]
    inner = pa.list_(pa.float32(), 1024) if fixed else pa.list_(pa.float32())
    emb = pa.array([[row.tolist() for row in d] for d in docs], type=pa.list_(inner))
    text = pa.array([[f"s{i}_{j}" for j in range(n)] for i, n in enumerate(lens)])
    table = pa.table({"text_sentences_sonar_emb": emb, "text_sentences": text})
    if split is None:
        pq.write_table(table, path, row_group_size=3)
    else:
        (path / f"split={split}").mkdir(parents=True, exist_ok=True)
        pq.write_table(table, path / f"split={split}" / "part-0.parquet", row_group_size=2)
    return docs


@pytest.mark.parametrize("fixed", [True, False])
def test_read_documents_and_batches(tmp_path, fixed):
    lens = [3, 1, 5, 2, 4, 2, 6]
    docs = _write(tmp_path / "d.parquet", lens, fixed)
    got, texts = read_sonar_documents(str(tmp_path / "d.parquet"), "text_sentences_sonar_emb", text_column="text_sentences")
    assert [g.shape for g in got] == [(n, 1024) for n in lens]
    assert all(np.array_equal(g.numpy(), d) for g, d in zip(got, docs))
    assert texts[2] == [f"s2_{j}" for j in range(5)]
    batches = list(iterate_prompt_batches(str(tmp_path / "d.parquet"), "text_sentences_sonar_emb", 3, device="cpu",
                                          max_prompt_len=4))
    assert [b.seqs.shape for b in batches] == [(3, 4, 1024), (3, 4, 1024), (1, 4, 1024)]
    assert batches[0].padding_mask.seq_lens.tolist() == [3, 1, 4]
    assert torch.equal(batches[0].seqs[1, 0], torch.from_numpy(docs[1][0])) and batches[0].seqs[1, 1:].abs().sum() == 0
    assert batches[2].padding_mask is None          # a batch whose documents have equal lengths carries no mask


def test_rank_shards_and_split_partition(tmp_path):
    root = tmp_path / "ds"
    train = _write(root, [2, 3, 4, 5, 6], True, seed=1, split="train")
    _write(root, [1, 1], True, seed=2, split="validation")
    parts = [read_sonar_documents(str(root), "text_sentences_sonar_emb", split="train", rank=r, world_size=2)[0] for r in range(2)]
    assert [len(p) for p in parts] == [3, 2]
    flat = parts[0] + parts[1]
    assert all(np.array_equal(g.numpy(), d) for g, d in zip(flat, train))
    val, _ = read_sonar_documents(str(root), "text_sentences_sonar_emb", split="validation")
    assert [v.shape[0] for v in val] == [1, 1]


def test_get_embeddings_sequence_with_targets():
    src = [torch.ones(2, 4), torch.ones(3, 4)]
    tgt = [2 * torch.ones(1, 4), 2 * torch.ones(2, 4)]
    b = get_embeddings_sequence(src, tgt)
    assert b.seqs.shape == (2, 5, 4) and b.source_lengths.tolist() == [2, 3]
    assert b.padding_mask.seq_lens.tolist() == [3, 5]
    d = get_embeddings_sequence(src, tgt, double_target=True)
    assert d.seqs.shape == (2, 7, 4) and d.padding_mask.seq_lens.tolist() == [4, 7]


def test_prompt_affixes(tmp_path):
    """Already-encoded prefix / suffix sentences around every prompt (the reference wraps sources in its
    source_prefix_text / source_suffix_text, lcm/datasets/dataloading.py:597-613)."""
    lens = [3, 1, 5]
    docs = _write(tmp_path / "d.parquet", lens, True)
    pre, suf = torch.full((2, 1024), 7.0), torch.full((1, 1024), -7.0)
    (b,) = list(iterate_prompt_batches(str(tmp_path / "d.parquet"), "text_sentences_sonar_emb", 8, device="cpu",
                                       max_prompt_len=4, prefix_embeddings=pre, suffix_embeddings=suf))
    assert b.seqs.shape == (3, 7, 1024) and b.padding_mask.seq_lens.tolist() == [6, 4, 7]
    assert torch.equal(b.seqs[1, :2], pre) and torch.equal(b.seqs[1, 2], torch.from_numpy(docs[1][0])) and torch.equal(b.seqs[1, 3:4], suf)
    assert torch.equal(b.seqs[2, 2:6], torch.from_numpy(docs[2][:4])) and torch.equal(b.seqs[2, 6:7], suf)
    with pytest.raises(ValueError, match="prefix_embeddings"):
        list(iterate_prompt_batches(str(tmp_path / "d.parquet"), "text_sentences_sonar_emb", 8, device="cpu",
                                    prefix_embeddings=torch.zeros(2, 512)))
