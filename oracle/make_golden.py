"""Writes tests/golden/toy_two_tower_golden.pt: a small seeded problem solved by the oracle
(inputs, noise, per-step latents of the first generated sentence, final outputs).

The reference publishes no numeric vectors for this path and cannot be imported here, so these
vectors do not pin the oracle against the reference; they pin the oracle against itself (any
later change of its arithmetic shows up) and give the GPU tests a fixture that needs no oracle
run.  Usage: python -m oracle.make_golden
"""

import os

import torch

import oracle
from oracle.weights import synthetic_noise, synthetic_prompts

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "toy_two_tower_golden.pt")
ARCH = "toy_two_tower_diffusion_lcm"
INIT = dict(seed=7, mod_gain=10.0, qk_gain=2.0)
OPTS = dict(guidance_scale=3.0, guidance_rescale=0.7, initial_noise_scale=0.6, inference_timesteps=10,
            eos_threshold=None)
B, P, G = 2, 3, 3


def main() -> None:
    cfg = oracle.get_arch(ARCH)
    sd = oracle.init_state_dict(cfg, **INIT)
    x = synthetic_prompts(oracle.load_dummy_normalizer(), B, P, seed=99)
    nz = synthetic_noise(G, B, seed=98)
    trace = []
    gen = oracle.OracleGenerator(oracle.OracleTwoTower(cfg, sd), oracle.GenOptions(**OPTS))
    hyps = gen(x, max_gen_len=G, noise=nz, trace=trace)
    steps = len(gen.model.scheduler.timesteps)
    first = trace[:steps]                                     # diffusion steps of the first generated sentence
    blob = {
        "arch": ARCH, "init": INIT, "options": OPTS, "B": B, "P": P, "G": G,
        "prompts": x, "noise": nz,
        "timesteps": gen.model.scheduler.timesteps.clone(),
        "latents_step0": first[0][1], "latents_step1": first[1][1], "latents_last": first[-1][1],
        "x0_last": first[-1][2],
        "outputs": torch.stack([h[0].seq for h in hyps]),     # (B, P+G, 1024)
    }
    torch.save(blob, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
