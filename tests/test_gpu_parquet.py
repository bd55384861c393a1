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


def test_generate_continuations_of_a_prompt_list(normalizer):
    """`lcm.inference.predict.generate_continuations`, the embedding-level part of the reference's predictor
    (lcm/evaluation/predictors/lcm.py:205-262): prompt cut, generator batches, padding, `max_gen_len_ratio`, results in
    prompt order without their prompts -- against the oracle, document by document (device-drawn noise is replaced by
    `noise=` per batch through the generator keyword, indexed by the batch's own documents)."""
    from lcm.inference.predict import generate_continuations
    from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    from oracle.weights import synthetic_prompts
    cfg = O.get_arch("mini_two_tower_diffusion_lcm")
    sd = O.init_state_dict(cfg, seed=4, mod_gain=10.0, qk_gain=2.0)
    lens = [5, 2, 3, 4, 1]
    full = synthetic_prompts(normalizer, len(lens), 6, seed=55)
    prompts = [full[i, :n + 1] for i, n in enumerate(lens)]                 # one sentence more than will be kept
    model = create_two_tower_diffusion_lcm_model(mini_config(), device="cuda:0", dtype=torch.float32)
    model.load_state_dict(sd)
    opts = dict(DEFAULT_OPTS, inference_timesteps=10, seed=9)
    gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**opts))
    outs = generate_continuations(gen, [p.cuda() for p in prompts], max_gen_len_ratio=0.5, max_prompt_len=5, generator_batch_size=2)
    kept = [min(n + 1, 5) for n in lens]
    # batches of (2, 2, 1) documents; each generates max(1, int(longest kept prompt * 0.5)) sentences
    want_len = [2, 2, 2, 2, 1]
    assert [o.shape for o in outs] == [(g, 1024) for g in want_len]
    # the same batches by hand, with the draws the seeded generator makes on the device, through the oracle
    om = O.OracleTwoTower(cfg, sd)
    for b0 in range(0, 5, 2):
        idx = list(range(b0, min(b0 + 2, 5)))
        G = want_len[b0]
        pl = [kept[i] for i in idx]
        x = torch.zeros(len(idx), max(pl), 1024)
        for r, i in enumerate(idx):
            x[r, :pl[r]] = prompts[i][:pl[r]]
        torch.manual_seed(9)
        steps = max(pl) - min(pl) + G
        nz = torch.zeros(steps, len(idx), 1024)
        for st in range(steps):          # one (active documents, 1, D) draw per step; a document leaves after its last sentence
            active = [r for r in range(len(idx)) if st < pl[r] - min(pl) + G]
            nz[st, active] = torch.randn(len(active), 1, 1024, device="cuda:0")[:, 0].cpu()
        ref = O.OracleGenerator(om, O.GenOptions(**{k: v for k, v in opts.items() if k != "seed"}))(
            x, prompt_lens=torch.tensor(pl), max_gen_len=G, noise=nz)
        for r, i in enumerate(idx):
            a, b = outs[i].cpu(), ref[r][0].seq[pl[r]:]
            an, bn = (a - normalizer["center"]) / normalizer["scale"], (b - normalizer["center"]) / normalizer["scale"]
            assert a.shape == b.shape and float((an - bn).abs().max()) <= FP32_MAX_ABS, (i, float((an - bn).abs().max()))
    del model
