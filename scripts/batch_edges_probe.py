"""Probe: 1.6B two-tower, bf16, batch sizes at the row-count boundaries of the product routing (32 / 64 / 128 / 256 /
384 rows = 2 x documents): every document's first generated sentence against the same document generated alone."""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
import oracle
from oracle.weights import synthetic_noise, synthetic_prompts
from parity_utils import DEFAULT_OPTS, run_hip
from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs

arch = "two_tower_diffusion_lcm_1_6B"
model = create_two_tower_diffusion_lcm_model(lcm_archs.get_config(arch), device="cuda:0", dtype=torch.bfloat16)
model.init_parameters(seed=3, modulator_std=0.02)
P, G, BMAX = 4, 2, 385
norm = oracle.load_dummy_normalizer()
x = synthetic_prompts(norm, BMAX, P, seed=21)
nz = synthetic_noise(P + G, BMAX, seed=22)
opts = dict(DEFAULT_OPTS, inference_timesteps=10)
c, s = norm["center"], norm["scale"]
alone = {}
def single(i):
    if i not in alone:
        o, _ = run_hip(arch, None, x[i:i + 1], nz[:, i:i + 1], G, opts, dtype=torch.bfloat16, model=model)
        alone[i] = (o.hypotheses[0][0].seq.float().cpu()[P:] - c) / s
    return alone[i]
for B in (1, 2, 3, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 96, 97, 127, 128, 129, 191, 192, 193, 255, 256, 257, 383, 384, 385):
    o, _ = run_hip(arch, None, x[:B], nz[:, :B], G, opts, dtype=torch.bfloat16, model=model)
    worst_cos, worst_abs = 1.0, 0.0
    for i in sorted({0, B // 2, B - 1}):
        v = (o.hypotheses[i][0].seq.float().cpu()[P:] - c) / s
        u = single(i)
        worst_cos = min(worst_cos, float(torch.nn.functional.cosine_similarity(u, v, dim=-1).min()))
        worst_abs = max(worst_abs, float((u - v).abs().max()))
    finite = all(torch.isfinite(h[0].seq.float()).all() for h in o.hypotheses)
    print(f"B={B:4d}: min cosine to the document alone {worst_cos:.6f}  max-abs {worst_abs:.3e}  finite {finite}", flush=True)
