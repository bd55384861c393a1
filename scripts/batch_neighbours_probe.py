"""Follow-up: per-document agreement between neighbouring batch sizes (every document of the smaller batch), first and
second generated sentence separately, 1.6B bf16."""
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
def batch(B):
    o, _ = run_hip(arch, None, x[:B], nz[:, :B], G, opts, dtype=torch.bfloat16, model=model)
    return torch.stack([(h[0].seq.float().cpu()[P:] - c) / s for h in o.hypotheses])      # (B, G, D)
outs = {B: batch(B) for B in (126, 127, 128, 191, 192, 193, 254, 255, 256, 257, 383, 384)}
for a, b in ((126, 128), (127, 128), (191, 192), (193, 192), (254, 256), (255, 256), (257, 256), (383, 384), (255, 257), (127, 255)):
    n = min(a, b)
    cs = torch.nn.functional.cosine_similarity(outs[a][:n], outs[b][:n], dim=-1)       # (n, G)
    m1, i1 = cs[:, 0].min(0)
    m2, i2 = cs[:, 1].min(0)
    print(f"B={a:3d} vs B={b:3d}: sentence 1 min cosine {float(m1):.6f} (doc {int(i1)}), median {float(cs[:, 0].median()):.6f};"
          f"  sentence 2 min {float(m2):.6f} (doc {int(i2)}), median {float(cs[:, 1].median()):.6f}", flush=True)
