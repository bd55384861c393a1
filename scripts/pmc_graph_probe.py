"""Probe (diagnostic): one sentence of the mini model through the graph-replayed sampling loop, with a chosen number
of diffusion steps, i.e. a chosen number of kernel nodes in the recorded graph.  Run under `rocprofv3 --pmc ...` with
LCMHIP_DEBUG_GRAPH=1 to see which graph API call a counter-collecting profiler stalls in, and from which graph size.

usage: python scripts/pmc_graph_probe.py <inference_timesteps> [sentences]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from lcm.datasets.batch import EmbeddingsBatch  # noqa: E402
from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator  # noqa: E402
from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model  # noqa: E402
from parity_utils import mini_config  # noqa: E402

steps = int(sys.argv[1])
sentences = int(sys.argv[2]) if len(sys.argv) > 2 else 2
model = create_two_tower_diffusion_lcm_model(mini_config(), device="cuda:0", dtype=torch.bfloat16, seed=0)
model.init_parameters(seed=0, modulator_std=0.2)
gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(
    guidance_scale=3.0, guidance_rescale=0.7, initial_noise_scale=0.6, inference_timesteps=steps, eos_threshold=None))
x = torch.randn(8, 4, 1024, device="cuda:0").to(torch.bfloat16) * 0.01
t0 = time.time()
with torch.cuda.stream(torch.cuda.Stream()):
    out = gen(EmbeddingsBatch(x, None), max_gen_len=sentences)
torch.cuda.synchronize()
n_exec = len(model.noise_scheduler.timesteps)
print(f"probe ok: {n_exec} executed steps, about {n_exec * 30 + 5} kernel nodes per graph, {sentences} sentences, "
      f"{time.time() - t0:.2f} s", flush=True)
