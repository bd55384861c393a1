"""Calibration sweep of the library's stream-copy kernel: walk pattern, cache policy, workgroups per CU, buffer size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm import _hip
dev = torch.device("cuda:0"); st = _hip.stream_ptr(dev)
def t(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters
for mb in (64, 256, 512, 1024):
    n = mb << 20
    a = torch.empty(n, dtype=torch.uint8, device=dev).random_(); b = torch.empty_like(a)
    us = t(lambda: b.copy_(a)); print(f"{mb:5d} MiB torch copy_             {2*n/us/1e3:8.1f} GB/s", flush=True)
    for per_cu in (4, 8, 16, 32):
        for m, name in ((0, "stride"), (1, "stride nt"), (2, "tiled"), (3, "tiled nt")):
            us = t(lambda: _hip.check(_hip.lib.lcmhip_stream_copy(_hip.ptr(a), _hip.ptr(b), n, m | (per_cu << 8), st)))
            print(f"{mb:5d} MiB {name:10s} {per_cu:2d} WG/CU    {2*n/us/1e3:8.1f} GB/s", flush=True)
    assert torch.equal(a, b)
