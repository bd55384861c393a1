"""Experiment: the sampling loop of 64 documents as ONE chain of 128-row kernels, or as 2 / 4 independent chains of
32 / 16 documents replayed concurrently from separate streams (each chain is its own session and graph; weights
shared).  Documents never interact, so the results are the same; what changes is how much of one chain's launch /
ramp / drain latency is covered by another chain's work.

usage: python scripts/two_chain_probe.py [--docs 64] [--gen 12] [--chains 1 2 4]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions  # noqa: E402
from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs  # noqa: E402
from lcm.models.two_tower_diffusion_lcm.builder import DeviceSession  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--docs", type=int, default=64)
ap.add_argument("--prompt", type=int, default=8)
ap.add_argument("--gen", type=int, default=12)
ap.add_argument("--start-context", type=int, default=64, help="context length at which the timed sentences start")
ap.add_argument("--chains", type=int, nargs="+", default=[1, 2, 4])
args = ap.parse_args()

dev = torch.device("cuda:0")
model = create_two_tower_diffusion_lcm_model(lcm_archs.get_config("two_tower_diffusion_lcm_1_6B"), device=dev,
                                             dtype=torch.bfloat16, seed=0)
model.init_parameters(seed=0, modulator_std=0.2)
model.prep_for_denoising(DiffusionLCMGeneratorOptions(guidance_scale=3.0, guidance_rescale=0.7, initial_noise_scale=0.6,
                                                      inference_timesteps=40))
B, P0, G = args.docs, args.start_context, args.gen
g = torch.Generator(device=dev).manual_seed(1)
prompt = torch.randn(B, P0, 1024, generator=g, device=dev) * 0.01
noise = torch.randn(G, B, 1024, generator=g, device=dev)
results = {}
outs = {}
for n in args.chains:
    assert B % n == 0
    b = B // n
    streams = [torch.cuda.Stream(dev) for _ in range(n)]
    sessions = []
    for i in range(n):
        with torch.cuda.stream(streams[i]):
            s = DeviceSession(model, b, P0 + G + 1, len(model.noise_scheduler.timesteps))
            s.set_schedule()
            sessions.append(s)

    def run():
        last = []
        for i, s in enumerate(sessions):
            with torch.cuda.stream(streams[i]):
                s.reset(b)
                s.append(prompt[i * b:(i + 1) * b], normalized=False)
                last.append(None)
        for step in range(G):
            for i, s in enumerate(sessions):          # enqueue every chain's sentence; they run concurrently on the GPU
                with torch.cuda.stream(streams[i]):
                    if last[i] is not None:
                        s.append(last[i].unsqueeze(1), normalized=False)
                    last[i] = s.denoise(noise[step, i * b:(i + 1) * b])
        torch.cuda.synchronize(dev)
        return torch.cat(last)

    torch.cuda.synchronize(dev)
    run()                                             # records the graphs
    t0 = time.perf_counter()
    out = run()
    dt = time.perf_counter() - t0
    outs[n] = out
    results[n] = dict(chains=n, docs_per_chain=b, seconds=round(dt, 4), ms_per_sentence_step=round(1e3 * dt / G, 3),
                      sentences_per_s=round(B * G / dt, 1))
    print(json.dumps(results[n]), flush=True)
    del sessions
base = outs[args.chains[0]]
for n in args.chains[1:]:
    c = torch.nn.functional.cosine_similarity(outs[n], base, dim=-1).min()
    print(f"chains={n}: min cosine of the last sentence against chains={args.chains[0]}: {float(c):.6f}")
