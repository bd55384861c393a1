"""The step before the path, on the GPU: documents stored in the reference's parquet layout (a list of 1024-d SONAR
vectors per row, lcm/datacards/datacards.yaml:1-20) -> `iterate_prompt_batches` -> `TwoTowerDiffusionLCMGenerator`,
compared with the oracle on the same documents; and the same documents sharded over two ranks give the same
sentences (noise is indexed by global document id)."""

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq
import pytest
import torch

import oracle as O
from parity_utils import DEFAULT_OPTS, compare, mini_config

pytestmark = pytest.mark.gpu

FP32_MAX_ABS, FP32_COS = 1e-3, 0.9999


def _write_dataset(path, lens, normalizer, seed=3):
    rng = np.random.default_rng(seed)
    c, s = normalizer["center"].numpy(), normalizer["scale"].numpy()
    docs = [(c + s * rng.standard_normal((n, 1024))).astype(np.float32) for n in lens]
    emb = pa.array([[row.tolist() for row in d] for d in docs], type=pa.list_(pa.list_(pa.float32(), 1024)))
    text = pa.array([[f"sentence {i}.{j}" for j in range(n)] for i, n in enumerate(lens)])
    pq.write_table(pa.table({"prompt_sentences_sonar_emb": emb, "prompt_sentences": text}), path, row_group_size=2)
    return [torch.from_numpy(d) for d in docs]


def test_parquet_prompts_through_the_generator(tmp_path, normalizer):
    from lcm.datasets.parquet_prompts import iterate_prompt_batches
    from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    from oracle.weights import synthetic_noise
    lens = [3, 1, 4, 2, 4, 3, 2]                                   # ragged; 7 documents -> batches of 3, 3, 1
    path = str(tmp_path / "prompts.parquet")
    docs = _write_dataset(path, lens, normalizer)
    cfg = O.get_arch("mini_two_tower_diffusion_lcm")
    sd = O.init_state_dict(cfg, seed=2, mod_gain=10.0, qk_gain=2.0)
    model = create_two_tower_diffusion_lcm_model(mini_config(), device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    opts = dict(DEFAULT_OPTS, inference_timesteps=10)
    G, P_max = 3, 4
    noise = synthetic_noise(P_max + G, len(lens), seed=8)           # [generation step, GLOBAL document]
    gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**opts))
    oracle_gen = O.OracleGenerator(O.OracleTwoTower(cfg, sd), O.GenOptions(**opts))

    def run(rank, world):
        """This rank's documents, in order: (global id, generated sequence)."""
        from lcm.utils.distributed import shard_bounds
        lo, _ = shard_bounds(len(lens), rank, world)
        out, first = [], lo
        for batch in iterate_prompt_batches(path, "prompt_sentences_sonar_emb", 3, device="cuda:0", rank=rank, world_size=world):
            n = batch.seqs.shape[0]
            ids = list(range(first, first + n))
            # noise rows of this batch's documents; the generator consumes noise[step - min prompt length]
            res = gen(batch, max_gen_len=G, noise=noise[:, ids].cuda())
            out += [(i, h[0].seq.float().cpu()) for i, h in zip(ids, res.hypotheses)]
            first += n
        return out

    got = dict(run(0, 1))
    assert sorted(got) == list(range(len(lens)))
    # the oracle on the same batches (same padding, same per-batch minimum prompt length)
    for b0 in range(0, len(lens), 3):
        ids = list(range(b0, min(b0 + 3, len(lens))))
        blens = [lens[i] for i in ids]
        x = torch.zeros(len(ids), max(blens), 1024)
        for r, i in enumerate(ids):
            x[r, :lens[i]] = docs[i]
        ref = oracle_gen(x, prompt_lens=torch.tensor(blens) if len(set(blens)) > 1 else None, max_gen_len=G, noise=noise[:, ids])

        class _H:
            def __init__(self, seq):
                self.seq = seq
        worst, cos = compare([[_H(got[i])] for i in ids], ref, sd)
        assert worst <= FP32_MAX_ABS and cos >= FP32_COS, (ids, worst, cos)
        for r, i in enumerate(ids):
            assert got[i].shape[0] == lens[i] + G and torch.equal(got[i][:lens[i]], docs[i])
    # two ranks: every document is generated exactly once, by the rank that owns it
    sharded = dict(run(0, 2) + run(1, 2))
    assert sorted(sharded) == sorted(got)
    assert all(sharded[i].shape[0] == lens[i] + G for i in sharded)
