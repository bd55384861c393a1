"""Diagnostic: per-diffusion-step divergence of the HIP path from the oracle (test infrastructure)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import oracle
from parity_utils import DEFAULT_OPTS, make_problem, _config
from lcm.datasets.batch import EmbeddingsBatch
from lcm.nn.incremental_state import LCMIncrementalStateBag
from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions
from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model

arch = sys.argv[1]
dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[2]]
kw = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    kw[k] = float(v)
B, P = 4, 3
cfg, sd, x, nz = make_problem(arch, B, P, 1, **kw)
if dtype == torch.bfloat16:
    sd = {k: v.bfloat16().float() for k, v in sd.items()}
    x = x.bfloat16().float()
opts = dict(DEFAULT_OPTS, inference_timesteps=40)
m = oracle.OracleTwoTower(cfg, sd)
g = oracle.OracleGenerator(m, oracle.GenOptions(**opts))
tr = []
g(x, max_gen_len=1, noise=nz, trace=tr)
model = create_two_tower_diffusion_lcm_model(_config(arch), device="cuda:0", dtype=dtype)
model.load_state_dict(sd)
model.prep_for_denoising(DiffusionLCMGeneratorOptions(**opts))
bag = LCMIncrementalStateBag(P + 1)
xc = x.cuda().to(dtype)
ctx = model.encode(EmbeddingsBatch(xc[:, :P - 1]), bag, normalized=False)
bag.increment_step_nr(P - 1)
steps = len(model.noise_scheduler.timesteps)
trace = torch.zeros(steps, B, 1024, device="cuda:0")
model.predict_next_sentence(EmbeddingsBatch(xc[:, P - 1:P]), ctx, context_state_bag=bag, noise=nz[0].cuda(), trace=trace)
lat = trace.cpu()
for i in range(steps):
    want = tr[i][1].reshape(B, 1024)
    d = (lat[i] - want).abs().max()
    c = torch.nn.functional.cosine_similarity(lat[i], want, dim=-1).min()
    if i < 6 or i % 8 == 0 or i == steps - 1:
        print(f"step {i:2d} t={int(tr[i][0]):3d} maxabs {float(d):.3e} mincos {float(c):.6f} |lat| {float(want.abs().max()):.3f}", flush=True)
