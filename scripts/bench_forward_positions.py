"""Training-shaped forward of the 1.6B two-tower diffusion LCM: contextualizer + denoiser over every position at once
(B documents x T positions, per-position diffusion timesteps, shifted causal cross-attention), bf16, one MI355X.
Prints one JSON line: forwards/s, positions/s, TFLOP/s against the dense bf16 matrix-core peak, per-kernel table."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm.datasets.batch import EmbeddingsBatch
from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs

MFMA_PEAK_TFLOPS = 2500.0
B, T = int(os.environ.get("FP_BATCH", "32")), int(os.environ.get("FP_SEQ", "128"))
dev = torch.device("cuda:0")
cfg = lcm_archs.get_config("two_tower_diffusion_lcm_1_6B")
model = create_two_tower_diffusion_lcm_model(cfg, device=dev, dtype=torch.bfloat16, seed=0)
model.init_parameters(seed=0, modulator_std=0.02)
g = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(B, T, 1024, generator=g, device=dev).to(torch.bfloat16)          # normalized clean embeddings
noisy = torch.randn(B, T, 1024, generator=g, device=dev)
ts = torch.randint(0, cfg.noise_scheduler.num_diffusion_train_steps, (B, T), generator=g, device=dev)
batch, noisy_batch = EmbeddingsBatch(x, None), EmbeddingsBatch(noisy, None, diffusion_timesteps=ts)

def step():
    return model.forward(batch, noisy_batch).seqs

for _ in range(2):
    out = step()
torch.cuda.synchronize()
K = 5
t0 = time.perf_counter()
for _ in range(K):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
assert torch.isfinite(out).all()
d = cfg.model_dim
Fc, Lc = cfg.context_encoder.allocated_ffn_dim(), cfg.context_encoder.num_layers
Fd, Ld = cfg.denoiser.allocated_ffn_dim(), cfg.denoiser.num_layers
rows = B * T
# contextualizer (q, k, v, o + SwiGLU) + cross K/V projections + denoiser (folded self-attention, cross q / o, SwiGLU)
gemm_flops = 2.0 * rows * (Lc * (4 * d * d + 3 * d * Fc) + 1024 * d + Ld * 2 * d * d
                           + Ld * (3 * d * d + 3 * d * Fd) + 2 * 1024 * d)
attn_flops = (Lc + Ld) * B * 4.0 * d * T * (T + 1) / 2
sess = model._session_pool[0]
sess.profile_enable(True)
step()
prof = {r["name"]: dict(launches=r["launches"], ms=round(r["total_ms"], 3),
                        tflops=round(r["flops"] / r["total_ms"] / 1e9, 1) if r["total_ms"] > 0 else None)
        for r in sess.profile() if r["launches"]}
sess.profile_enable(False)
print(json.dumps({"workload": f"two_tower_diffusion_lcm_1_6B forward over all positions, {B} x {T}, bf16",
                  "ms_per_forward": round(dt * 1e3, 2), "positions_per_s": round(rows / dt, 1),
                  "tflops": round((gemm_flops + attn_flops) / dt / 1e12, 1),
                  "frac_of_mfma_peak": round((gemm_flops + attn_flops) / dt / 1e12 / MFMA_PEAK_TFLOPS, 3),
                  "gemm_tflop": round(gemm_flops / 1e12, 2), "attention_tflop": round(attn_flops / 1e12, 3), "kernels": prof}))
