"""Experiment: one batch of 64 documents as K concurrent sub-batches (threads + streams) on one GPU."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from lcm.datasets.batch import EmbeddingsBatch
from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator
from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs

dev = torch.device("cuda:0")
B, P, G = 64, 8, int(os.environ.get("G", "16"))
model = create_two_tower_diffusion_lcm_model(lcm_archs.get_config(bench.ARCH), device=dev, dtype=torch.bfloat16, seed=0)
model.init_parameters(seed=0, modulator_std=0.2)
opts = dict(bench.OPTS, max_seq_len=max(G, bench.OPTS["max_seq_len"]))
prompts = bench.make_prompts(model, B, P, 0, dev).to(model.dtype)
noise = bench.make_noise(G, B, 0, dev)

def run(K):
    gens = [TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**opts)) for _ in range(K)]
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    per = B // K
    outs = [None] * K
    def work(i):
      try:
        with torch.cuda.stream(streams[i]):
            outs[i] = gens[i](EmbeddingsBatch(prompts[i * per:(i + 1) * per], None), max_gen_len=G,
                              noise=noise[:, i * per:(i + 1) * per].contiguous())
      except Exception as e:
        import traceback; traceback.print_exc()
    def once():
        ts = [threading.Thread(target=work, args=(i,)) for i in range(K)]
        for t in ts: t.start()
        for t in ts: t.join()
        torch.cuda.synchronize()
    once()
    t0 = time.perf_counter(); once(); dt = time.perf_counter() - t0
    print(f"K={K}: {B * G / dt:8.1f} sentence-embeddings/s  ({dt * 1e3:.0f} ms)", flush=True)
    return torch.stack([h[0].seq for o in outs for h in o.hypotheses]).float()

a = run(1)
for K in (2, 4):
    b = run(K)
    cos = torch.nn.functional.cosine_similarity(a[:, P:], b[:, P:], dim=-1).min()
    print(f"   min cosine vs K=1: {float(cos):.5f}")
