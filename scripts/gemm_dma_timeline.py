"""Where do the microseconds of a weight-streaming product go?  Runs with a library built with -DLCM_DMA_STAMPS
(LCMHIP_LIB=scripts/alt/liblcm_hip_stamps.so): every workgroup of an unsplit LDS-DMA product writes four 100 MHz
wall-clock stamps (start, first K chunk landed, K loop done, epilogue done) into the split-K workspace.  Products of the
sampling loop's shapes are launched back to back over 13 different weight sets (cold weights, as in the loop)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm import _hip as hip

dev = torch.device("cuda:0")
torch.manual_seed(0)
M = int(os.environ.get("TL_ROWS", "128"))
st = torch.cuda.current_stream().cuda_stream


def run(name, N, K, epi, dual):
    sets = 13
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(sets)]
    W2 = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(sets)] if dual else [None] * sets
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.zeros(1 << 20, device=dev, dtype=torch.int64)
    def launch(i):
        hip.check(hip.lib.lcmhip_gemm(hip.ptr(A), K, hip.ptr(W[i]), hip.ptr(W2[i]), None, None, None, hip.ptr(out), N, M, N, K,
                                      epi, hip.BF16, hip.ptr(ws), 0 if dual else 0, st))
    # an unsplit launch: a zero-byte workspace keeps the K range whole while the pointer carries the stamps
    for rep in range(3):
        for i in range(sets):
            launch(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(sets):
        launch(i)
    e1.record()
    torch.cuda.synchronize()
    n_wg = (2 * N if dual else N) // 64 * ((M + 127) // 128 if M > 64 else 1)
    t = ws[:4 * n_wg].view(n_wg, 4).cpu().double() * 0.01      # microseconds
    if not bool((t > 0).all()):                                # a library without stamps: the launch time only
        print(f"{name}: {M} x {N} x {K}, {e0.elapsed_time(e1) * 1e3 / sets:.2f} us per launch back to back")
        return
    t0 = t[:, 0].min()
    rel = t - t0
    q = lambda v: f"min {v.min():5.2f} median {v.median():5.2f} max {v.max():5.2f}"   # noqa: E731
    print(f"{name}: {M} x {N} x {K}, {n_wg} workgroups, {e0.elapsed_time(e1) * 1e3 / sets:.2f} us per launch back to back")
    print(f"  workgroup start after the first one   {q(rel[:, 0])}")
    print(f"  start -> first chunk landed           {q(t[:, 1] - t[:, 0])}")
    print(f"  first chunk -> K loop done            {q(t[:, 2] - t[:, 1])}")
    print(f"  epilogue                              {q(t[:, 3] - t[:, 2])}")
    print(f"  last workgroup done after first start {rel[:, 3].max():5.2f}")


run("SwiGLU up-projection", 5632, 2048, hip.EPI_SWIGLU, True)
run("d x d, unsplit", 2048, 2048, hip.EPI_STORE, False)
run("down-projection, unsplit", 2048, 5632, hip.EPI_STORE, False)
