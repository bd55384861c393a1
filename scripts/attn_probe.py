"""Cross-attention kernel probe: 64 documents x 16 heads, one query each, `keys` cached positions (cap 137), bf16.
Times the launch over 13 rotating K/V caches (cold, like the 13 denoiser layers) and over one cache (hot: the K/V
stay in the 256 MB Infinity Cache), for several key counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm import _hip

B, H, cap = int(os.environ.get("AP_DOCS", "64")), 16, 137
dev = torch.device("cuda:0")
st = torch.cuda.Stream(dev)
with torch.cuda.stream(st):
    caches = [(torch.randn(B, H, cap, 128, device=dev).bfloat16(), torch.randn(B, H, cap, 128, device=dev).bfloat16()) for _ in range(13)]
    q = torch.randn(B, H * 128, device=dev).bfloat16()
    out = torch.empty_like(q)
    sp = _hip.stream_ptr(dev)

    def run(sets, keys, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for i in range(n):
            k, v = sets[i % len(sets)]
            _hip.check(_hip.lib.lcmhip_attention(_hip.ptr(q), _hip.ptr(k), _hip.ptr(v), None, _hip.ptr(out), B, 1, H, cap, keys, 0,
                                                 _hip.BF16, sp))
        e1.record(st)
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    for keys in (9, 33, 73, 137):
        run(caches, keys, 26)
        cold = run(caches, keys, 260)
        hot = run(caches[:1], keys, 260)
        mb = 2 * B * H * keys * 256 / 1e6
        print(f"keys {keys:4d}  {mb:6.1f} MB   cold {cold:6.2f} us ({mb / cold:.2f} TB/s)   hot {hot:6.2f} us", flush=True)
