"""Microbenchmark of lcmhip_gemm / lcmhip_gemm_partial at the denoiser's shapes (cold weights: 13 rotating sets)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm import _hip
dev = torch.device("cuda:0")
SETS, ITERS = int(os.environ.get("GB_SETS", "13")), 40
st = _hip.stream_ptr(dev)

def bench(name, M, N, K, mode):
    PAD = int(os.environ.get("GB_PAD", "0"))     # extra elements per activation row (spreads rows over L2 channels)
    A = torch.randn(M, K + PAD, device=dev).bfloat16()
    Ws = [torch.randn(N, K, device=dev).bfloat16() * 0.02 for _ in range(SETS)]
    W2s = [torch.randn(N, K, device=dev).bfloat16() * 0.02 for _ in range(SETS)] if mode == "swiglu" else None
    slab = torch.empty(12 * M * N, device=dev, dtype=torch.float32)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    splits = C.c_int32(0)
    def run(i):
        if mode == "swiglu":
            _hip.check(_hip.lib.lcmhip_gemm(_hip.ptr(A), K + PAD, _hip.ptr(Ws[i]), _hip.ptr(W2s[i]), None, None, None, _hip.ptr(out), N,
                                            M, N, K, _hip.EPI_SWIGLU, _hip.BF16, None, 0, st))
        else:
            _hip.check(_hip.lib.lcmhip_gemm_partial(_hip.ptr(A), K + PAD, _hip.ptr(Ws[i]), _hip.ptr(slab), slab.numel() * 4,
                                                    C.byref(splits), M, N, K, _hip.BF16, st))
    for i in range(SETS):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(ITERS):
        for i in range(SETS):
            run(i)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / (ITERS * SETS)
    wbytes = N * K * 2 * (2 if mode == "swiglu" else 1)
    print(f"{name:10s} M={M:4d} N={N:5d} K={K:5d} splits={splits.value:2d}  {us:7.2f} us  {wbytes / us / 1e6:6.2f} TB/s(weights)", flush=True)

bench("ffn_in", 128, 5632, 2048, "swiglu")
bench("ffn_out", 128, 2048, 5632, "partial")
bench("self_ov", 128, 2048, 2048, "partial")
bench("cross_q", 64, 2048, 2048, "partial")
