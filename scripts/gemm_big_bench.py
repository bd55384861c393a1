"""TFLOP/s of lcmhip_gemm at many-row shapes (bf16, random operands, hot caches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm import _hip
dev = torch.device("cuda:0")
st = _hip.stream_ptr(dev)
ITERS = int(os.environ.get("GB_ITERS", "20"))
NO_TORCH = bool(os.environ.get("GB_NO_TORCH"))       # under a counter-collecting profiler: only this library's kernels
ROWS = [int(v) for v in os.environ.get("GB_ROWS", "256,1024,4096").split(",")]
# split-K workspace handed to the library, as a session does for every linear product (MiB; 0: none)
WS = torch.empty(int(os.environ.get("GB_WS_MB", "96")) << 18, device=dev, dtype=torch.float32) if int(os.environ.get("GB_WS_MB", "96")) else None

def bench(M, N, K, epi):
    A = torch.randn(M, K, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
    W2 = (torch.randn(N, K, device=dev) * 0.02).bfloat16() if epi == _hip.EPI_SWIGLU else None
    f32_out = epi in (_hip.EPI_RESID, _hip.EPI_STORE_F32)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32_out else torch.bfloat16)
    def run():
        _hip.check(_hip.lib.lcmhip_gemm(_hip.ptr(A), K, _hip.ptr(W), _hip.ptr(W2), None, None, None, _hip.ptr(out), N,
                                        M, N, K, epi, _hip.BF16, _hip.ptr(WS), WS.numel() * 4 if WS is not None else 0, st))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS): run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / ITERS
    flops = 2.0 * M * N * K * (2 if epi == _hip.EPI_SWIGLU else 1)
    us_t = float("inf")
    if not NO_TORCH:
        torch.nn.functional.linear(A, W)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(ITERS): torch.nn.functional.linear(A, W)
        e1.record(); torch.cuda.synchronize()
        us_t = 1e3 * e0.elapsed_time(e1) / ITERS
    print(f"M={M:5d} N={N:5d} K={K:5d} epi={epi}  {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s   (torch/hipBLASLt linear: {2.0*M*N*K / us_t / 1e6:7.1f})", flush=True)

def ab_n128():
    """The 256 x 128 tile against the 128 x 128 kernel on the one-tile-per-CU shapes, alternating inside one process."""
    shapes = [(4096, 2048, 2048, _hip.EPI_STORE), (4096, 2048, 5632, _hip.EPI_STORE), (4096, 2048, 2048, _hip.EPI_RESID),
              (4096, 2048, 5632, _hip.EPI_RESID), (4096, 6144, 2048, _hip.EPI_STORE), (2048, 2048, 2048, _hip.EPI_STORE)]
    for M, N, K, epi in shapes:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == _hip.EPI_RESID else torch.bfloat16)
        res = {}
        for rnd in range(3):
            for mode in ("0", "1"):
                os.environ["LCMHIP_BIG256_N128"] = mode
                def run():
                    _hip.check(_hip.lib.lcmhip_gemm(_hip.ptr(A), K, _hip.ptr(W), None, None, None, None, _hip.ptr(out), N,
                                                    M, N, K, epi, _hip.BF16, None, 0, st))
                for _ in range(3): run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50): run()
                e1.record(); torch.cuda.synchronize()
                res.setdefault(mode, []).append(1e3 * e0.elapsed_time(e1) / 50)
        a, b = min(res["0"]), min(res["1"])
        print(f"M={M} N={N} K={K} epi={epi}: 128x128 kernel {a:7.1f} us, 256x128 tile {b:7.1f} us  ({100 * (a / b - 1):+.1f} %)", flush=True)
    os.environ.pop("LCMHIP_BIG256_N128", None)

def ab_env(var, on, shapes, label_off, label_on, iters=50):
    """One environment switch of the library (read at every launch), alternating inside one process."""
    for M, N, K, epi in shapes:
        A = torch.randn(M, K, device=dev).bfloat16()
        W = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        W2 = (torch.randn(N, K, device=dev) * 0.02).bfloat16() if epi == _hip.EPI_SWIGLU else None
        out = torch.zeros(M, N, device=dev, dtype=torch.float32 if epi == _hip.EPI_RESID else torch.bfloat16)
        res = {}
        for rnd in range(3):
            for mode in ("0", on):
                os.environ[var] = mode
                def run():
                    _hip.check(_hip.lib.lcmhip_gemm(_hip.ptr(A), K, _hip.ptr(W), _hip.ptr(W2), None, None, None, _hip.ptr(out), N,
                                                    M, N, K, epi, _hip.BF16, None, 0, st))
                for _ in range(3): run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters): run()
                e1.record(); torch.cuda.synchronize()
                res.setdefault(mode, []).append(1e3 * e0.elapsed_time(e1) / iters)
        a, b = min(res["0"]), min(res[on])
        fl = 2.0 * M * N * K * (2 if epi == _hip.EPI_SWIGLU else 1)
        print(f"M={M} N={N} K={K} epi={epi}: {label_off} {a:7.1f} us ({fl / a / 1e6:6.0f} TF), {label_on} {b:7.1f} us ({fl / b / 1e6:6.0f} TF)  ({100 * (a / b - 1):+.1f} %)", flush=True)
    os.environ.pop(var, None)

if os.environ.get("GB_AB_N128"):
    pr = torch.cuda.get_device_properties(0)
    print(f"device: {pr.name}, {pr.multi_processor_count} CUs, clock {getattr(pr, 'clock_rate', 0) / 1e3:.0f} MHz, "
          f"memory clock {getattr(pr, 'memory_clock_rate', 0) / 1e3:.0f} MHz, {pr.total_memory / 2**30:.0f} GiB", flush=True)
    ab_n128()
    sys.exit(0)

for M in ROWS:
    bench(M, 2048, 2048, _hip.EPI_STORE)
    bench(M, 5632, 2048, _hip.EPI_SWIGLU)
    bench(M, 2048, 5632, _hip.EPI_STORE)
    if M >= 1024:
        bench(M, 2048, 2048, _hip.EPI_RESID)
        bench(M, 2048, 5632, _hip.EPI_RESID)
        bench(M, 6144, 2048, _hip.EPI_STORE)
bench(4096, 4096, 4096, _hip.EPI_STORE)
bench(8192, 8192, 8192, _hip.EPI_STORE)
