"""Diagnostic (test infrastructure): where a bf16 HIP run leaves the bf16-emulating oracle.
Prints, for one sentence of the given arch: the contextualizer output against the emulation (fraction of elements
that differ and by how many bf16 steps), then the divergence of the latents per diffusion step against the
emulation and against the fp32 oracle.

usage: python scripts/diag_emu.py <arch> [mod_gain=.. qk_gain=..]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import oracle  # noqa: E402
from oracle.model import KVCache  # noqa: E402
from parity_utils import DEFAULT_OPTS, _config, make_problem  # noqa: E402
from lcm.datasets.batch import EmbeddingsBatch  # noqa: E402
from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions  # noqa: E402
from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model  # noqa: E402
from lcm.nn.incremental_state import LCMIncrementalStateBag  # noqa: E402

arch = sys.argv[1]
kw = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    kw[k] = float(v)
B, P = 4, 5
cfg, sd, x, nz = make_problem(arch, B, P, 1, **kw)
x16 = oracle.r16(x)
sd16 = {k: oracle.r16(v) for k, v in sd.items()}
opts = dict(DEFAULT_OPTS, inference_timesteps=40)
go = oracle.GenOptions(**opts)


def run_oracle(model):
    model.prep_for_denoising(go)
    cache = KVCache(P + 1)
    with torch.inference_mode():
        ctx = model.encode(model.normalize(x16[:, :P - 1]), cache)
        cache.increment_step_nr(P - 1)
        tr = []
        out, ctx = model.predict_next_sentence(x16[:, P - 1:P], ctx, cache, nz[0].unsqueeze(1), trace=tr)
    return ctx, tr, out


ctx_emu, tr_emu, out_emu = run_oracle(oracle.OracleTwoTowerBf16(cfg, sd))
ctx_ref, tr_ref, out_ref = run_oracle(oracle.OracleTwoTower(cfg, sd16))

model = create_two_tower_diffusion_lcm_model(_config(arch), device="cuda:0", dtype=torch.bfloat16)
model.load_state_dict(sd)
from lcm import _hip  # noqa: E402
model.session_flags = _hip.SESSION_KEEP_CONTEXT      # the contextualizer output is read back below
model.prep_for_denoising(DiffusionLCMGeneratorOptions(**opts))
bag = LCMIncrementalStateBag(P + 1)
xc = x.cuda().to(torch.bfloat16)
ctx = model.encode(EmbeddingsBatch(xc[:, :P - 1]), bag, normalized=False)
bag.increment_step_nr(P - 1)
steps = len(model.noise_scheduler.timesteps)
trace = torch.zeros(steps, B, 1024, device="cuda:0")
out, ctx = model.predict_next_sentence(EmbeddingsBatch(xc[:, P - 1:P]), ctx, context_state_bag=bag, noise=nz[0].cuda(),
                                       trace=trace)
torch.cuda.synchronize()
hip_ctx = ctx.seqs.float().cpu()                    # (B, P, d) bf16 values
d = (hip_ctx - ctx_emu)
ulp = torch.maximum(ctx_emu.abs(), torch.tensor(1e-30)).log2().floor().exp2() * 2.0 ** -7     # bf16 spacing at the value
rel = (d.abs() / ulp)
print(f"context vs emulation: differing elements {float((d != 0).float().mean()):.5f}, "
      f"max difference {float(rel.max()):.1f} bf16 steps, > 1 step: {float((rel > 1.5).float().mean()):.6f}")
dr = hip_ctx - ctx_ref
print(f"context vs fp32 oracle: rms rel diff {float(dr.pow(2).mean().sqrt() / ctx_ref.pow(2).mean().sqrt()):.5f}; "
      f"emulation vs fp32 oracle: {float((ctx_emu - ctx_ref).pow(2).mean().sqrt() / ctx_ref.pow(2).mean().sqrt()):.5f}")
for pos in range(P):
    dd = hip_ctx[:, pos] - ctx_emu[:, pos]
    print(f"  position {pos}: differing {float((dd != 0).float().mean()):.5f}")

lat = trace.cpu()
cosf = torch.nn.functional.cosine_similarity
for i in range(steps):
    we, wr = tr_emu[i][1].reshape(B, 1024), tr_ref[i][1].reshape(B, 1024)
    if i < 8 or i % 6 == 0 or i == steps - 1:
        n = float(we.pow(2).mean().sqrt())
        print(f"step {i:2d} t={int(tr_emu[i][0]):3d}  HIP-emu rms {float((lat[i] - we).pow(2).mean().sqrt()) / n:.3e} "
              f"cos {float(cosf(lat[i], we, dim=-1).min()):.6f} | HIP-fp32 rms {float((lat[i] - wr).pow(2).mean().sqrt()) / n:.3e} "
              f"cos {float(cosf(lat[i], wr, dim=-1).min()):.6f} | emu-fp32 rms {float((we - wr).pow(2).mean().sqrt()) / n:.3e}",
              flush=True)
