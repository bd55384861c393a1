"""Which call of ANOTHER host thread breaks a thread-local stream capture on this runtime?  Thread A records a small graph
over and over (capture_error_mode thread_local); thread B meanwhile repeats one kind of call on a stream of its own."""
import threading, time
import torch

dev = torch.device("cuda", 0)
a = torch.randn(1 << 20, device=dev)
torch.cuda.synchronize()


def b_ops():
    s = torch.cuda.Stream(dev)
    big = torch.randn(1 << 20, device=dev)
    host = torch.randn(1 << 16)
    pinned = torch.randn(1 << 16).pin_memory()
    ev = torch.cuda.Event()
    return s, {
        "nothing": lambda: time.sleep(0.0005),
        "kernel on own stream": lambda: big.mul_(1.0001),
        "stream.synchronize": lambda: s.synchronize(),
        "event record + synchronize": lambda: (ev.record(s), ev.synchronize()),
        "D2H copy (.cpu())": lambda: big[:1024].cpu(),
        "H2D copy from pageable": lambda: host.to(dev),
        "H2D copy from pinned, non_blocking": lambda: pinned.to(dev, non_blocking=True),
        "fresh allocation (hipMalloc)": lambda: (torch.cuda.empty_cache(), torch.empty(1 << 22, device=dev)),
        "torch.cuda.synchronize (device)": lambda: torch.cuda.synchronize(),
        "stream create": lambda: torch.cuda.Stream(dev),
    }


def trial(name):
    stop = threading.Event()
    fails, caps, berr = [], [0], []

    def thread_a():
        torch.cuda.set_device(0)
        sa = torch.cuda.Stream(dev)
        out = torch.empty_like(a)
        while not stop.is_set():
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, stream=sa, capture_error_mode="thread_local"):
                    for _ in range(20):
                        torch.mul(a, 2.0, out=out)
                caps[0] += 1
            except Exception as e:  # noqa: BLE001
                fails.append(repr(e)[:160])
                torch.cuda.synchronize()

    def thread_b():
        torch.cuda.set_device(0)
        s, ops = b_ops()
        with torch.cuda.stream(s):
            while not stop.is_set():
                try:
                    ops[name]()
                except Exception as e:  # noqa: BLE001
                    berr.append(repr(e)[:160])
                    break

    ta, tb = threading.Thread(target=thread_a), threading.Thread(target=thread_b)
    ta.start(); tb.start()
    time.sleep(1.5)
    stop.set()
    ta.join(); tb.join()
    torch.cuda.synchronize()
    print(f"{name:40s} captures ok {caps[0]:5d}  failed {len(fails):4d}  {fails[0] if fails else ''}  {'B: ' + berr[0] if berr else ''}", flush=True)


for n in b_ops()[1]:
    trial(n)
