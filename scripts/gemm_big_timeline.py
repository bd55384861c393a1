"""Timeline of the 256 x 128 many-row tile: a library built with -DLCM_DMA_STAMPS (LCMHIP_LIB=scripts/alt/liblcm_hip_stamps.so)
lets every workgroup write four 100 MHz wall-clock stamps (start, first K tile landed, K loop done, epilogue done)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcm import _hip as hip

dev = torch.device("cuda:0")
torch.manual_seed(0)
st = torch.cuda.current_stream().cuda_stream


def run(name, M, N, K, epi):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = [torch.randn(N, K, device=dev).to(torch.bfloat16) for _ in range(4)]
    f32 = epi in (hip.EPI_RESID, hip.EPI_STORE_F32)
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    ws = torch.zeros(1 << 20, device=dev, dtype=torch.int64)
    def launch(i):
        hip.check(hip.lib.lcmhip_gemm(hip.ptr(A), K, hip.ptr(W[i % 4]), None, None, None, None, hip.ptr(out), N, M, N, K,
                                      epi, hip.BF16, hip.ptr(ws), 0, st))
    for i in range(8):
        launch(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(8):
        launch(i)
    e1.record()
    torch.cuda.synchronize()
    n_wg = (M // 256) * (N // 128)
    t = ws[:4 * n_wg].view(n_wg, 4).cpu().double() * 0.01      # microseconds
    print(f"{name}: {M} x {N} x {K}, {n_wg} workgroups, {e0.elapsed_time(e1) * 1e3 / 8:.2f} us per launch back to back")
    if not bool((t > 0).all()):
        return
    t0 = t[:, 0].min()
    order = torch.argsort(t[:, 0])
    q = lambda v: f"min {v.min():6.2f} median {v.median():6.2f} max {v.max():6.2f}"   # noqa: E731
    for rnd in range((n_wg + 255) // 256):
        sel = order[rnd * 256:(rnd + 1) * 256]
        r = t[sel]
        print(f"  round {rnd} ({len(sel)} workgroups by start time): start {q(r[:, 0] - t0)}")
        print(f"      start -> first K tile   {q(r[:, 1] - r[:, 0])}")
        print(f"      K loop                  {q(r[:, 2] - r[:, 1])}")
        print(f"      epilogue                {q(r[:, 3] - r[:, 2])}")
        print(f"      done                    {q(r[:, 3] - t0)}")


run("attention output (RESID)", 4096, 2048, 2048, hip.EPI_RESID)
run("plain store", 4096, 2048, 2048, hip.EPI_STORE)
run("q / k / v width, plain store", 4096, 6144, 2048, hip.EPI_STORE)
run("down-projection (RESID)", 4096, 2048, 5632, hip.EPI_RESID)
