"""Base-LCM benchmark configuration: base_lcm 1.6B next-concept regression forward, batch 32 x 128 concepts, bf16, one MI355X.
Prints one JSON line: forwards/s, concepts/s, TFLOP/s against the dense bf16 matrix-core peak, per-kernel table."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm.datasets.batch import EmbeddingsBatch
from lcm.models.base_lcm import create_base_lcm_model, lcm_archs
from lcm.utils.card_utils import load_model_from_card

MFMA_PEAK_TFLOPS = 2500.0
B, S = int(os.environ.get("BF_BATCH", "32")), int(os.environ.get("BF_SEQ", "128"))
dev = torch.device("cuda:0")
cfg = lcm_archs.get_config("base_lcm_1_6B")
model = create_base_lcm_model(cfg, device=dev, dtype=torch.bfloat16, seed=0)
model.sonar_normalizer.load_state_dict(load_model_from_card("dummy_sonar_normalizer", device=dev, dtype=torch.bfloat16).state_dict())
c, s = model.sonar_normalizer.center.float(), model.sonar_normalizer.scale.float()
g = torch.Generator(device=dev).manual_seed(1234)
x = (c + s * torch.randn(B, S, 1024, generator=g, device=dev)).to(torch.bfloat16)

def step():
    return model.predict_next_sentence(EmbeddingsBatch(x, None)).seqs

def measure(K):
    for _ in range(2):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        out = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K, out

# BF_AB=1: the stacked q / k / v product against three separate ones; BF_AB=rope: the rotation / cache scatter in the
# stacked product's epilogue against the separate pass.  Alternating in one process (a session takes the model's
# session_flags when it is created).
if os.environ.get("BF_AB"):
    from lcm import _hip
    flag = _hip.SESSION_SEPARATE_ROPE if os.environ["BF_AB"] == "rope" else _hip.SESSION_SEPARATE_QKV
    for rnd in range(3):
        for mode in (0, flag):
            model.session_flags = mode
            dt, _ = measure(10)
            print(f"round {rnd} session_flags={mode}: {dt * 1e3:.2f} ms per forward", flush=True)
    model.session_flags = 0
# BF_AB_N128=1: the 256 x 128 tile for the one-tile-per-CU products (attn_out, ffn_out, stacked q / k / v) against the
# 128 x 128 kernel, alternating in one process (the library reads LCMHIP_BIG256_N128 at every launch)
if os.environ.get("BF_AB_N128"):
    for rnd in range(3):
        for mode in ("0", "1"):
            os.environ["LCMHIP_BIG256_N128"] = mode
            dt, _ = measure(10)
            print(f"round {rnd} LCMHIP_BIG256_N128={mode}: {dt * 1e3:.2f} ms per forward", flush=True)
    os.environ["LCMHIP_BIG256_N128"] = os.environ.get("BF_N128_FINAL", "1")
# BF_AB_ENVVAR=NAME: any switch the library reads at every launch, "0" against "1", alternating in one process
if os.environ.get("BF_AB_ENVVAR"):
    name = os.environ["BF_AB_ENVVAR"]
    for rnd in range(3):
        for mode in ("0", "1"):
            os.environ[name] = mode
            dt, _ = measure(10)
            print(f"round {rnd} {name}={mode}: {dt * 1e3:.2f} ms per forward", flush=True)
    os.environ.pop(name, None)
dt, out = measure(int(os.environ.get("BF_STEPS", "10")))
assert torch.isfinite(out).all()
d, F, L, H = cfg.model_dim, cfg.lcm.allocated_ffn_dim(), cfg.lcm.num_layers, cfg.lcm.num_attn_heads
rows = B * S
gemm_flops = 2.0 * rows * (L * (4 * d * d + 3 * d * F) + 2 * 1024 * d)
attn_flops = L * B * 4.0 * d * S * (S + 1) / 2
sess = model._session_pool[0]
sess.profile_enable(True)
step()
prof = {r["name"]: dict(launches=r["launches"], ms=round(r["total_ms"], 3),
                        tflops=round(r["flops"] / r["total_ms"] / 1e9, 1) if r["total_ms"] > 0 else None)
        for r in sess.profile() if r["launches"]}
sess.profile_enable(False)
print(json.dumps({"workload": f"base_lcm_1_6B forward, {B} x {S} concepts, bf16", "ms_per_forward": round(dt * 1e3, 2),
                  "concepts_per_s": round(rows / dt, 1), "tflops": round((gemm_flops + attn_flops) / dt / 1e12, 1),
                  "frac_of_mfma_peak": round((gemm_flops + attn_flops) / dt / 1e12 / MFMA_PEAK_TFLOPS, 3),
                  "gemm_tflop": round(gemm_flops / 1e12, 2), "attention_tflop": round(attn_flops / 1e12, 3), "kernels": prof}))
