"""Race screen for the 256 x 128 tile kernel (hand-issued LDS reads, counted waits, LDS-DMA in flight across barriers):
random many-row shapes, each computed by the 128 x 128 kernel (LCMHIP_BIG256_N128=0) and by the 256 x 128 tile kernel, both
wave layouts, and compared bit for bit (every output sums its K tiles in the same order in all of them).  A second
stream keeps the memory system busy with copies meanwhile.  usage: python scripts/soak_n128.py [rounds=300]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm import _hip

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
st = _hip.stream_ptr(dev)
rng = random.Random(1234)
noise_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
noise_b = torch.empty_like(noise_a)
side = torch.cuda.Stream(dev)
bad = 0
for it in range(rounds):
    M = rng.choice([384, 512, 1000, 2048, 3000, 4096, 5000, 8192]) + rng.choice([0, 0, 1, 37])
    N = rng.choice([1024, 2048, 4096, 6144])
    K = rng.choice([256, 512, 1024, 2048, 5632])
    epi = rng.choice([_hip.EPI_STORE, _hip.EPI_RESID, _hip.EPI_STORE_F32, _hip.EPI_RELU])
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    b = torch.randn(N, device=dev).bfloat16()
    f32 = epi in (_hip.EPI_RESID, _hip.EPI_STORE_F32)
    res0 = torch.randn(M, N, device=dev) if f32 else None
    outs = []
    with torch.cuda.stream(side):
        noise_b.copy_(noise_a, non_blocking=True)
    for mode in ("0", "1"):
        os.environ["LCMHIP_BIG256_N128"] = mode
        out = res0.clone() if f32 else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        _hip.check(_hip.lib.lcmhip_gemm(_hip.ptr(A), K, _hip.ptr(W), None, _hip.ptr(b), None, None, _hip.ptr(out), N, M, N, K, epi,
                                        _hip.BF16, None, 0, st))
        outs.append(out)
    torch.cuda.synchronize()
    for o in outs[1:]:
        if not torch.equal(o, outs[0]):
            bad += 1
            print(f"MISMATCH round {it}: M={M} N={N} K={K} epi={epi}: {(o.float() - outs[0].float()).abs().max().item()}", flush=True)
    if it % 50 == 49:
        print(f"{it + 1} rounds, {bad} mismatches", flush=True)
print(f"done: {rounds} rounds, {bad} mismatches")
sys.exit(1 if bad else 0)
