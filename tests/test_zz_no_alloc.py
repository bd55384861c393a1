"""lcmhip_gemm_static, the entry point that allocates nothing: its split-K slab is a static device array instead of
a caller workspace.  Checked through the C ABI against lcmhip_gemm given a workspace of the same size (identical
split plan, so bit for bit), on two calls in a row, the second on a smaller problem; the static array keeps the first
call's contents in between.  Collected last (file name) so that a failure here hides no other test."""

import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from lcm import _hip
    return _hip


def _call(hip, static, A, W, W2, bias, bias2, gate, out, epi, ws):
    code = hip.dtype_code(A.dtype)
    M, K = A.shape
    N = W.shape[0]
    st = hip.stream_ptr()
    if static:
        return hip.lib.lcmhip_gemm_static(hip.ptr(A), A.stride(0), hip.ptr(W), hip.ptr(W2), hip.ptr(bias), hip.ptr(bias2),
                                          hip.ptr(gate), hip.ptr(out), out.stride(0), M, N, K, epi, code, st)
    return hip.lib.lcmhip_gemm(hip.ptr(A), A.stride(0), hip.ptr(W), hip.ptr(W2), hip.ptr(bias), hip.ptr(bias2),
                               hip.ptr(gate), hip.ptr(out), out.stride(0), M, N, K, epi, code, hip.ptr(ws),
                               ws.numel() * 4, st)


def _problem(dtype, M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    A = (torch.randn(M, K, generator=g, device="cuda") * 0.5).to(dtype)
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.05).to(dtype)
    b = (torch.randn(N, generator=g, device="cuda") * 0.1).to(dtype)
    gate = torch.randn(N, generator=g, device="cuda")
    resid = torch.randn(M, N, generator=g, device="cuda")
    return A, W, b, gate, resid


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gemm_static_equals_gemm_with_workspace(hip, dtype):
    ws = torch.empty(hip.STATIC_SCRATCH_BYTES // 4, dtype=torch.float32, device="cuda")
    # two calls in a row on the shared static array: the sampling loop's split-K shape, then a smaller one
    for M, N, K, seed in [(128, 2048, 2048, 1), (48, 1024, 1024, 2)]:
        A, W, b, gate, resid = _problem(dtype, M, N, K, seed)
        ref32 = A.float() @ W.float().t() + b.float()
        for epi in (hip.EPI_STORE, hip.EPI_STORE_F32, hip.EPI_RESID):
            odt = dtype if epi == hip.EPI_STORE else torch.float32
            want = resid.clone() if epi == hip.EPI_RESID else torch.full((M, N), 7.0, device="cuda").to(odt)
            got = want.clone()
            g = gate if epi == hip.EPI_RESID else None
            ws.fill_(float("nan"))     # the workspace variant must not depend on its contents either
            hip.check(_call(hip, False, A, W, None, b, None, g, want, epi, ws))
            hip.check(_call(hip, True, A, W, None, b, None, g, got, epi, None))
            torch.cuda.synchronize()
            assert torch.equal(got, want), (dtype, M, N, K, epi, float((got.float() - want.float()).abs().max()))
            # and both are the product: fp32 exact to accumulation order, bf16 to operand rounding
            full = resid + gate * ref32 if epi == hip.EPI_RESID else ref32
            tol = 2e-2 if (dtype == torch.bfloat16 and epi == hip.EPI_STORE) else 2e-3
            assert (got.float() - full).abs().max() <= tol * max(1.0, float(full.abs().max())), (dtype, epi)


def test_gemm_static_allocates_nothing_and_runs_under_capture(hip):
    """What the entry point is for: (1) after lcmhip_static_init() (outside any capture) a call changes the device's
    free memory by nothing, warmed or not; (2) a call recorded into a HIP graph (stream capture forbids allocation) and
    replayed gives the bits of the eager call."""
    hip.check(hip.lib.lcmhip_static_init())
    hip.check(hip.lib.lcmhip_static_init())                     # idempotent
    M, N, K = 128, 2048, 2048
    A, W, b, _, _ = _problem(torch.bfloat16, M, N, K, 11)
    eager = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(3):
        hip.check(_call(hip, True, A, W, None, b, None, None, eager, hip.EPI_STORE, None))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0, (free0, free1)
    # under capture, on a side stream: the launches are recorded (nothing runs), then the graph is replayed
    captured = torch.zeros_like(eager)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        graph.capture_begin()
        rc = _call(hip, True, A, W, None, b, None, None, captured, hip.EPI_STORE, None)
        graph.capture_end()
    hip.check(rc)
    torch.cuda.synchronize()
    assert not captured.any()                                    # capture executes nothing
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(captured, eager)
    captured.zero_()
    graph.replay()                                               # the static array keeps the first replay's contents
    torch.cuda.synchronize()
    assert torch.equal(captured, eager)
    del graph


def test_gemm_static_swiglu_and_limits(hip):
    A, W, b, _, _ = _problem(torch.bfloat16, 64, 512, 1024, 3)
    _, W2, b2, _, _ = _problem(torch.bfloat16, 64, 512, 1024, 4)
    ws = torch.empty(hip.STATIC_SCRATCH_BYTES // 4, dtype=torch.float32, device="cuda")
    want = torch.empty(64, 512, dtype=torch.bfloat16, device="cuda")
    got = torch.empty_like(want)
    hip.check(_call(hip, False, A, W, W2, b, b2, None, want, hip.EPI_SWIGLU, ws))
    hip.check(_call(hip, True, A, W, W2, b, b2, None, got, hip.EPI_SWIGLU, None))
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    # beyond the static maximum: a clear error, no launch
    M, N = 2048, 2048        # 4 Mi outputs > LCMHIP_STATIC_MAX_OUTPUTS (2 Mi)
    A = torch.zeros(M, 256, dtype=torch.bfloat16, device="cuda")
    W = torch.zeros(N, 256, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    rc = _call(hip, True, A, W, None, None, None, None, out, hip.EPI_STORE, None)
    assert rc == -1 and b"LCMHIP_STATIC_MAX_OUTPUTS" in hip.lib.lcmhip_last_error()
