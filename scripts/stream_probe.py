"""How fast can a one-shot kernel stream N cold megabytes?  torch reductions over rotating buffers."""
import torch
dev = torch.device("cuda:0")
for mb in (8, 23, 46, 92, 368):
    n = mb * 1024 * 1024 // 2
    sets = max(2, 700 // mb)
    bufs = [torch.randn(n, device=dev, dtype=torch.bfloat16) for _ in range(sets)]
    outs = torch.empty(sets, device=dev)
    def run():
        for i, b in enumerate(bufs):
            torch.amax(b.view(torch.int16), dim=0, out=outs[i:i+1].view(torch.int16)[:1]) if False else b.view(torch.int32).max()
    run(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 10
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / (iters * sets)
    print(f"{mb:4d} MB cold int32 max: {us:7.1f} us per launch  {mb * 1.048576 / us:6.2f} TB/s", flush=True)
    del bufs
