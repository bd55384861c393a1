"""HBM roofline microbenchmark of the fused elementwise kernels at many rows (>= 64k), where they are
bandwidth- rather than launch-bound.  Prints achieved GB/s against the 8 TB/s peak."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm import _hip
dev = torch.device("cuda:0")
st = _hip.stream_ptr(dev)
PEAK = 8000.0
ROWS = int(os.environ.get("EB_ROWS", "131072"))
res = {}

def timeit(fn, nbytes, name, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    gbs = nbytes / us / 1e3
    res[name] = dict(us=round(us, 1), GBps=round(gbs, 1), frac=round(gbs / PEAK, 3), bytes=nbytes)
    print(f"{name:34s} {us:9.1f} us  {gbs:8.1f} GB/s  {gbs / PEAK:5.3f} of peak", flush=True)

D = 1024
# calibration: what plain device-to-device copies of the same sizes achieve on this part
for dt, es in ((torch.bfloat16, 2), (torch.float32, 4)):
    a = torch.randn(ROWS, D, device=dev).to(dt); b = torch.empty_like(a)
    timeit(lambda: b.copy_(a), 2 * ROWS * D * es, f"torch copy_ {dt}".replace("torch.", "", 1))
    for nt in (0, 1):       # the library's own calibration copy: 16 B per lane, 16-KB tiles per workgroup, 16 workgroups per CU
        timeit(lambda: _hip.check(_hip.lib.lcmhip_stream_copy(_hip.ptr(a), _hip.ptr(b), a.numel() * es, 2 | nt | (16 << 8), st)),
               2 * ROWS * D * es, f"stream_copy {'nt ' if nt else ''}{dt}".replace("torch.", ""))
for dt, code, es in ((torch.bfloat16, _hip.BF16, 2), (torch.float32, _hip.F32, 4)):
    x = torch.randn(ROWS, D, device=dev).to(dt); y = torch.empty_like(x)
    c = torch.randn(D, device=dev).to(dt); s = (torch.rand(D, device=dev) + 0.5).to(dt)
    timeit(lambda: _hip.check(_hip.lib.lcmhip_normalize(_hip.ptr(x), _hip.ptr(c), _hip.ptr(s), None, None, _hip.ptr(y), ROWS, D, code, st)),
           2 * ROWS * D * es, f"normalize {dt}".replace("torch.", ""))
    timeit(lambda: _hip.check(_hip.lib.lcmhip_denormalize(_hip.ptr(x), _hip.ptr(c), _hip.ptr(s), None, None, _hip.ptr(y), ROWS, D, code, st)),
           2 * ROWS * D * es, f"denormalize {dt}".replace("torch.", ""))
for dim in (1024, 2048):
    rows = ROWS * 1024 // dim
    x = torch.randn(rows, dim, device=dev); y = torch.empty(rows, dim, device=dev, dtype=torch.bfloat16)
    w = torch.ones(dim, device=dev).bfloat16(); sh = torch.randn(dim, device=dev); sc = torch.randn(dim, device=dev)
    timeit(lambda: _hip.check(_hip.lib.lcmhip_norm_modulate(_hip.ptr(x), _hip.ptr(w), None, _hip.ptr(sh), _hip.ptr(sc), _hip.ptr(y),
                                                            rows, dim, _hip.NORM_RMS, 1e-5, _hip.BF16, st)),
           rows * dim * (4 + 2), f"rmsnorm+modulate d={dim} f32->bf16")
x = torch.randn(ROWS, D, device=dev); oc = torch.randn(ROWS, D, device=dev); ou = torch.randn(ROWS, D, device=dev)
prev = torch.empty_like(x); x0 = torch.empty_like(x)
par = _hip.DenoiseParams(3.0, 0.7, 0.0, 0.0, 100.0, 0.6, 0, 0, 1.0, 0, 0.995, 1.0)
timeit(lambda: _hip.check(_hip.lib.lcmhip_ddim_cfg_step(_hip.ptr(x), _hip.ptr(oc), _hip.ptr(ou), _hip.ptr(prev), _hip.ptr(x0), ROWS, D,
                                                         0.5, 0.6, C.byref(par), None, st)),
       5 * ROWS * D * 4, "ddim+cfg step f32 (3 in, 2 out)")
timeit(lambda: _hip.check(_hip.lib.lcmhip_ddim_cfg_step(_hip.ptr(x), _hip.ptr(oc), _hip.ptr(ou), _hip.ptr(prev), None, ROWS, D,
                                                         0.5, 0.6, C.byref(par), None, st)),
       4 * ROWS * D * 4, "ddim+cfg step f32 (3 in, 1 out)")
print(json.dumps({"rows": ROWS, "peak_GBps": PEAK, "kernels": res}))
