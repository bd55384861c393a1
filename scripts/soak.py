"""Leak / stability soak: repeated generator calls with changing batch sizes; prints free device memory after each
round (a library-side leak shows as a monotonic drop) and checks that equal inputs give bit-equal outputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm.datasets.batch import EmbeddingsBatch
from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator
from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs

dev = torch.device("cuda:0")
model = create_two_tower_diffusion_lcm_model(lcm_archs.get_config("two_tower_diffusion_lcm_1_6B"), device=dev, dtype=torch.bfloat16, seed=0)
model.init_parameters(seed=0, modulator_std=0.2)
gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(guidance_scale=3.0, guidance_rescale=0.7, initial_noise_scale=0.6,
                                                                     inference_timesteps=40, eos_threshold=None, max_seq_len=16))
first = {}
for rnd in range(12):
    B = (64, 3, 17, 64)[rnd % 4]
    g = torch.Generator(device=dev).manual_seed(B)
    x = (0.01 * torch.randn(B, 8, 1024, generator=g, device=dev)).to(torch.bfloat16)
    nz = torch.randn(4, B, 1024, generator=g, device=dev)
    out = gen(EmbeddingsBatch(x, None), max_gen_len=4, noise=nz)
    seqs = torch.stack([h[0].seq for h in out.hypotheses]).float()
    assert torch.isfinite(seqs).all()
    if B in first:
        assert torch.equal(first[B], seqs), f"round {rnd}: outputs differ from the first run at batch {B}"
    else:
        first[B] = seqs.clone()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    print(f"round {rnd:2d} batch {B:3d}  free {free / 2**30:8.3f} GiB  torch reserved {torch.cuda.memory_reserved() / 2**30:6.3f} GiB", flush=True)
print("soak ok")
