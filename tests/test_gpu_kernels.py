"""Single-kernel parity through the C ABI: each HIP kernel against the oracle's statement of the
same operation (fp32: max-abs tolerances written at each check; bf16: against the fp32 result
of the bf16-rounded inputs)."""

import ctypes as C
import functools
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    assert torch.cuda.is_available()
    from lcm import _hip
    return _hip


def _st(hip):
    return hip.stream_ptr(torch.device("cuda:0"))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows", [1, 7, 300, 70001])
def test_normalizer_roundtrip(hip, normalizer, dtype, rows):
    torch.manual_seed(0)
    c, s = normalizer["center"].cuda().to(dtype), normalizer["scale"].cuda().to(dtype)
    x = (normalizer["center"] + normalizer["scale"] * torch.randn(rows, 1024)).cuda().to(dtype)
    y = torch.empty_like(x)
    hip.check(hip.lib.lcmhip_normalize(hip.ptr(x), hip.ptr(c), hip.ptr(s), None, None, hip.ptr(y), rows, 1024,
                                       hip.dtype_code(dtype), _st(hip)))
    ref = (x.float() - c.float()) / s.float()
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (y.float() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    z = torch.empty_like(x)
    hip.check(hip.lib.lcmhip_denormalize(hip.ptr(y), hip.ptr(c), hip.ptr(s), None, None, hip.ptr(z), rows, 1024,
                                         hip.dtype_code(dtype), _st(hip)))
    ref_z = y.float() * s.float() + c.float()
    assert (z.float() - ref_z).abs().max() <= (1e-6 if dtype == torch.float32 else 5e-4)


def test_normalizer_clip(hip, normalizer):
    c, s = normalizer["center"].cuda(), normalizer["scale"].cuda()
    x = (normalizer["center"] + 3 * normalizer["scale"] * torch.randn(5, 1024)).cuda()
    lo, hi = torch.full((1024,), -1.5).cuda(), torch.full((1024,), 2.0).cuda()
    y = torch.empty_like(x)
    hip.check(hip.lib.lcmhip_normalize(hip.ptr(x), hip.ptr(c), hip.ptr(s), hip.ptr(lo), hip.ptr(hi), hip.ptr(y), 5, 1024, 0, _st(hip)))
    ref = torch.clamp((x - c) / s, min=lo, max=hi)
    assert (y - ref).abs().max() <= 1e-5
    z = torch.empty_like(x)
    big = 3 * torch.randn(5, 1024).cuda()
    hip.check(hip.lib.lcmhip_denormalize(hip.ptr(big), hip.ptr(c), hip.ptr(s), hip.ptr(lo), hip.ptr(hi), hip.ptr(z), 5, 1024, 0, _st(hip)))
    assert (z - (torch.clamp(big, min=lo, max=hi) * s + c)).abs().max() <= 1e-7


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("kind", ["rms", "standard"])
@pytest.mark.parametrize("dim", [256, 1024, 2048])
@pytest.mark.parametrize("rows", [9, 1501])     # >= 1024 rows without a pending product: one wave per row
def test_norm_modulate(hip, dtype, kind, dim, rows):
    from oracle.model import norm as oracle_norm
    torch.manual_seed(1)
    x = (torch.randn(rows, dim) * 3 + 0.5).cuda()
    w = (1 + 0.1 * torch.randn(dim)).cuda().to(dtype)
    b = (0.1 * torch.randn(dim)).cuda().to(dtype) if kind == "standard" else None
    shift, scale = (0.3 * torch.randn(dim)).cuda(), (0.3 * torch.randn(dim)).cuda()
    sd = {"n.weight": w.float().cpu()}
    if b is not None:
        sd["n.bias"] = b.float().cpu()
    for mod in (False, True):
        y = torch.empty(rows, dim, device="cuda", dtype=dtype)
        hip.check(hip.lib.lcmhip_norm_modulate(hip.ptr(x), hip.ptr(w), hip.ptr(b), hip.ptr(shift) if mod else None,
                                               hip.ptr(scale) if mod else None, hip.ptr(y), rows, dim,
                                               1 if kind == "rms" else 0, 1e-5, hip.dtype_code(dtype), _st(hip)))
        ref = oracle_norm(x.cpu(), sd, "n", kind, 1e-5)
        if mod:
            ref = ref * (1 + scale.cpu()) + shift.cpu()
        tol = 2e-5 if dtype == torch.float32 else 4e-2
        assert (y.float().cpu() - ref).abs().max() <= tol, (kind, dim, mod)


def _gemm(hip, A, W, W2, bias, bias2, gate, out, epi, ws=None):
    M, K = A.shape
    N = W.shape[0]
    hip.check(hip.lib.lcmhip_gemm(hip.ptr(A), A.stride(0), hip.ptr(W), hip.ptr(W2), hip.ptr(bias), hip.ptr(bias2),
                                  hip.ptr(gate), hip.ptr(out), out.stride(0), M, N, K, epi, hip.dtype_code(A.dtype),
                                  hip.ptr(ws), 0 if ws is None else ws.numel() * 4, _st(hip)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M", [1, 17, 50, 64, 100, 128, 200])
@pytest.mark.parametrize("N,K", [(256, 256), (1024, 2048), (768, 256), (2048, 5632)])
@pytest.mark.parametrize("split", [False, True])
def test_gemm_epilogues(hip, dtype, M, N, K, split):
    _check_epilogues(hip, dtype, M, N, K, split)


@pytest.mark.parametrize("M", [256, 300, 384, 400, 1000, 2000])
@pytest.mark.parametrize("N,K", [(256, 256), (2048, 2048), (1024, 5632)])
@pytest.mark.parametrize("split", [False, True])
def test_gemm_many_rows(hip, M, N, K, split):
    """Several row tiles in bf16: up to 383 rows on the weight-streaming LDS-DMA tile, from 384 rows on the
    128 x 128 matrix-core tile kernel (when its tiles x splits cover the chip)."""
    _check_epilogues(hip, torch.bfloat16, M, N, K, split)


@pytest.mark.parametrize("M", [2049, 2304, 3000])
@pytest.mark.parametrize("K", [256, 768, 2048])
def test_gemm_one_tile_per_cu(hip, M, K):
    """225 tiles of 256 x 256 (M = 2049, ragged against the 256-row tile; 2304: exact) fit one wave of the 8-wave
    kernel whose LDS-DMA stays in flight across barriers, which is then chosen over two waves of 128 x 128 tiles;
    K = 256 is the shortest pipeline the interface allows (4 K tiles).  M = 3000 (300 tiles, two waves) goes to the
    128 x 128 kernel."""
    _check_epilogues(hip, torch.bfloat16, M, 6400, K, False)


@pytest.mark.parametrize("M,N,K", [(4096, 2048, 2048), (4000, 2048, 768), (3700, 6144, 256), (513, 1280, 5632)])
def test_gemm_256x128_tile(hip, monkeypatch, M, N, K):
    """The 256 x 128 tile kernel (the default routing of linear many-row products without a K split: wave tile 64 x 64,
    hand-issued fragment reads with counted waits) against float64 through every linear epilogue, and bit for bit
    against the 128 x 128 kernel (LCMHIP_BIG256_N128=0): every output sums its K tiles in the same order in both
    kernels.  Ragged row counts, the shortest K loop the interface allows (4 K tiles), several waves of tiles."""
    dt = torch.bfloat16
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K).cuda().to(dt)
    W = (torch.randn(N, K) / math.sqrt(K)).cuda().to(dt)
    b = torch.randn(N).cuda().to(dt)
    gate = torch.randn(N).cuda()
    res0 = torch.randn(M, N).cuda()

    def run_all():
        outs = []
        for epi in (hip.EPI_STORE, hip.EPI_RELU):
            o = torch.empty(M, N, device="cuda", dtype=dt)
            _gemm(hip, A, W, None, b, None, None, o, epi)
            outs.append(o)
        o32 = torch.empty(M, N, device="cuda")
        _gemm(hip, A, W, None, None, None, None, o32, hip.EPI_STORE_F32)
        outs.append(o32)
        r = res0.clone()
        _gemm(hip, A, W, None, b, None, gate, r, hip.EPI_RESID)
        outs.append(r)
        torch.cuda.synchronize()
        return outs

    monkeypatch.setenv("LCMHIP_BIG256_N128", "0")
    base = run_all()
    monkeypatch.setenv("LCMHIP_BIG256_N128", "1")
    got = run_all()
    lin = A.double().cpu() @ W.double().cpu().T
    ref = lin + b.double().cpu()
    scale = max(1.0, float(ref.abs().max()))
    assert (got[0].double().cpu() - ref).abs().max() <= 3e-2 * scale
    assert (got[2].double().cpu() - lin).abs().max() <= 2e-3 * scale
    assert (got[3].double().cpu() - (res0.double().cpu() + gate.double().cpu() * ref)).abs().max() <= 2e-3 * max(1.0, float((gate.double().cpu() * ref).abs().max()))
    for g, o in zip(got, base):
        assert torch.equal(g, o)


@pytest.mark.parametrize("M,N,K,epi", [(4096, 5632, 2048, "swiglu"), (1024, 5632, 2048, "swiglu"), (2600, 4096, 512, "store"),
                                         (4096, 4096, 1024, "resid")])
def test_gemm_256x256_tile(hip, monkeypatch, M, N, K, epi):
    """The 256 x 256 tile kernel (one workgroup per CU; taken where its waves of tiles finish first: the SwiGLU
    up-projection of many rows, large square products) bit for bit against the 128 x 128 kernel that LCMHIP_BIG256=0 and
    LCMHIP_BIG256_N128=0 route the same call to (the same K order per output), and against float64."""
    dt = torch.bfloat16
    torch.manual_seed(M + N + K)
    A = torch.randn(M, K).cuda().to(dt)
    W = (torch.randn(N, K) / math.sqrt(K)).cuda().to(dt)
    W2 = (torch.randn(N, K) / math.sqrt(K)).cuda().to(dt) if epi == "swiglu" else None
    b = torch.randn(N).cuda().to(dt)
    b2 = torch.randn(N).cuda().to(dt) if epi == "swiglu" else None
    gate = torch.randn(N).cuda() if epi == "resid" else None
    res0 = torch.randn(M, N).cuda()
    code = {"swiglu": hip.EPI_SWIGLU, "store": hip.EPI_STORE, "resid": hip.EPI_RESID}[epi]

    def run():
        o = res0.clone() if epi == "resid" else torch.empty(M, N, device="cuda", dtype=dt)
        _gemm(hip, A, W, W2, b, b2, gate, o, code)
        torch.cuda.synchronize()
        return o

    monkeypatch.setenv("LCMHIP_BIG256", "0")
    monkeypatch.setenv("LCMHIP_BIG256_N128", "0")
    base = run()
    monkeypatch.delenv("LCMHIP_BIG256")
    monkeypatch.delenv("LCMHIP_BIG256_N128")
    got = run()
    assert torch.equal(got, base)
    lin = A.double().cpu() @ W.double().cpu().T + b.double().cpu()
    if epi == "swiglu":
        ref = (A.double().cpu() @ W2.double().cpu().T + b2.double().cpu()) * torch.nn.functional.silu(lin)
    elif epi == "resid":
        ref = res0.double().cpu() + gate.double().cpu() * lin
    else:
        ref = lin
    tol = 2e-3 if epi == "resid" else 3e-2
    assert (got.double().cpu() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M", [65, 96, 128])
@pytest.mark.parametrize("N,K", [(5632, 2048), (6144, 256), (64, 256), (48 * 16, 768)])
def test_gemm_swiglu_guarded_window(hip, M, N, K):
    """SwiGLU with 65..128 bf16 rows (the sampling loop's up-projection) into a row-strided window of a guard
    buffer: rows past M and columns past N must stay untouched; K = 256 is the shortest pipeline (2 chunks)."""
    torch.manual_seed(N + K + M)
    A = torch.randn(M, K).cuda().to(torch.bfloat16)
    W = (torch.randn(N, K) / math.sqrt(K)).cuda().to(torch.bfloat16)
    W2 = (torch.randn(N, K) / math.sqrt(K)).cuda().to(torch.bfloat16)
    b = torch.randn(N).cuda().to(torch.bfloat16)
    b2 = torch.randn(N).cuda().to(torch.bfloat16)
    guard = torch.full((M + 3, N + 16), 7.0, device="cuda", dtype=torch.bfloat16)
    import ctypes
    # the output is a row-strided window of the guard buffer (ldo = N + 16)
    hip.check(hip.lib.lcmhip_gemm(hip.ptr(A), K, hip.ptr(W), hip.ptr(W2), hip.ptr(b), hip.ptr(b2), hip.ptr(None),
                                  ctypes.c_void_p(guard.data_ptr()), guard.stride(0), M, N, K, hip.EPI_SWIGLU,
                                  hip.dtype_code(A.dtype), hip.ptr(None), 0, _st(hip)))
    gate = A.double().cpu() @ W.double().cpu().T + b.double().cpu()
    ref = (A.double().cpu() @ W2.double().cpu().T + b2.double().cpu()) * torch.nn.functional.silu(gate)
    got = guard.double().cpu()
    assert (got[:M, :N] - ref).abs().max() <= 3e-2 * max(1.0, float(ref.abs().max()))
    assert (got[M:] == 7.0).all() and (got[:, N:] == 7.0).all()


def test_gemm_one_tile_per_cu_identity(hip):
    """A = I against an asymmetric integer W (exact in bf16): catches a wrong fragment, quadrant or half-tile map
    of the 256 x 256 kernel exactly, and a stale or early LDS read as a wrong integer."""
    M = K = N = 4096
    A = torch.eye(M, K).cuda().to(torch.bfloat16)
    W = (torch.arange(N).view(N, 1) % 7 - 3 + 2 * (torch.arange(K).view(1, K) % 5)).float().cuda().to(torch.bfloat16)
    want = W.float().cpu().T
    for _ in range(3):
        out = torch.full((M, N), float("nan"), device="cuda")
        _gemm(hip, A, W, None, None, None, None, out, hip.EPI_STORE_F32)
        assert torch.equal(out.cpu(), want)


def test_gemm_many_rows_identity_and_row_map(hip):
    """A = I against an asymmetric integer W (exact in bf16) through the 128 x 128 kernel, plain and scattered
    through a row map into the (slot, head, position, 128) cache layout."""
    M = K = 2048
    N = 2048
    A = torch.eye(M, K).cuda().to(torch.bfloat16)
    W = (torch.arange(N).view(N, 1) % 7 - 3 + 2 * (torch.arange(K).view(1, K) % 5)).float().cuda().to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda")
    _gemm(hip, A, W, None, None, None, None, out, hip.EPI_STORE_F32)
    assert torch.equal(out.cpu(), W.float().cpu().T)


def _check_epilogues(hip, dtype, M, N, K, split):
    torch.manual_seed(M * 1000 + N + K)
    ws = torch.empty(12 * M * N, device="cuda") if split else None
    _gemm = functools.partial(globals()["_gemm"], ws=ws)   # same calls, optional split-K scratch

    A = torch.randn(M, K).cuda().to(dtype)
    W = (torch.randn(N, K) / math.sqrt(K)).cuda().to(dtype)
    W2 = (torch.randn(N, K) / math.sqrt(K)).cuda().to(dtype)
    b = torch.randn(N).cuda().to(dtype)
    b2 = torch.randn(N).cuda().to(dtype)
    gate = torch.randn(N).cuda()
    Af, Wf, W2f, bf, b2f = (t.double().cpu() for t in (A, W, W2, b, b2))
    lin = Af @ Wf.T + bf
    tol = 2e-4 if dtype == torch.float32 else 3e-2       # bf16: output rounding of O(1) values
    # STORE (asymmetric data catches a transposed fragment map)
    out = torch.empty(M, N, device="cuda", dtype=dtype)
    _gemm(hip, A, W, None, b, None, None, out, hip.EPI_STORE)
    assert (out.double().cpu() - lin).abs().max() <= tol * max(1.0, float(lin.abs().max()))
    # STORE_F32 without bias
    out32 = torch.empty(M, N, device="cuda")
    _gemm(hip, A, W, None, None, None, None, out32, hip.EPI_STORE_F32)
    assert (out32.double().cpu() - (lin - bf)).abs().max() <= (2e-4 if dtype == torch.float32 else 2e-3) * max(1.0, float(lin.abs().max()))
    # RELU
    _gemm(hip, A, W, None, b, None, None, out, hip.EPI_RELU)
    assert (out.double().cpu() - lin.clamp(min=0)).abs().max() <= tol * max(1.0, float(lin.abs().max()))
    # SWIGLU: (inner) * silu(gate)
    _gemm(hip, A, W, W2, b, b2, None, out, hip.EPI_SWIGLU)
    ref = (Af @ W2f.T + b2f) * torch.nn.functional.silu(lin)
    assert (out.double().cpu() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    # RESID with and without gate
    res0 = torch.randn(M, N).cuda()
    res = res0.clone()
    _gemm(hip, A, W, None, b, None, gate, res, hip.EPI_RESID)
    ref = res0.double().cpu() + gate.double().cpu() * lin
    assert (res.double().cpu() - ref).abs().max() <= (2e-4 if dtype == torch.float32 else 2e-3) * max(1.0, float(ref.abs().max()))
    res = res0.clone()
    _gemm(hip, A, W, None, None, None, None, res, hip.EPI_RESID)
    ref = res0.double().cpu() + (lin - bf)
    assert (res.double().cpu() - ref).abs().max() <= (2e-4 if dtype == torch.float32 else 2e-3) * max(1.0, float(ref.abs().max()))


def test_gemm_identity_asymmetric(hip):
    """A = I against an asymmetric integer W: exact in bf16, exposes any row/col swap."""
    K = N = 256
    A = torch.eye(64, K).cuda().to(torch.bfloat16)
    W = (torch.arange(N).view(N, 1) % 7 - 3 + 2 * (torch.arange(K).view(1, K) % 5)).float().cuda().to(torch.bfloat16)
    out = torch.empty(64, N, device="cuda")
    _gemm(hip, A, W, None, None, None, None, out, hip.EPI_STORE_F32)
    assert torch.equal(out.cpu(), W.float().cpu().T[:64])


def test_gemm_rejects_bad_shapes(hip):
    A = torch.zeros(4, 100, device="cuda")
    W = torch.zeros(64, 100, device="cuda")
    out = torch.zeros(4, 64, device="cuda")
    with pytest.raises(RuntimeError, match="512-byte"):
        _gemm(hip, A, W, None, None, None, None, out, hip.EPI_STORE)


@pytest.mark.parametrize("thresholding", [False, True])
@pytest.mark.parametrize("eta", [0.0, 0.6])
@pytest.mark.parametrize("mode", ["plain_sample", "plain_epsilon", "plain_v", "cfg"])
def test_ddim_cfg_step(hip, mode, eta, thresholding):
    """One fused update against the oracle's scheduler: every prediction type, guidance, stochastic DDIM
    (ddim.py:473-506, host-supplied variance noise) and dynamic thresholding (ddim.py:242-280, torch.quantile)."""
    import oracle
    from oracle.archs import SchedulerConfig
    torch.manual_seed(3)
    rows, dim = 5, 1024
    ptype = {"plain_sample": "sample", "plain_epsilon": "epsilon", "plain_v": "v_prediction", "cfg": "sample"}[mode]
    cfg = SchedulerConfig(num_train_steps=100, prediction_type=ptype, thresholding=thresholding,
                          dynamic_thresholding_ratio=0.9, sample_max_value=1.7)
    sch = oracle.DDIMSchedule(cfg)
    sch.set_timesteps(40)
    t = 57
    x = torch.randn(rows, 1, dim) * 0.8
    oc, ou = torch.randn(rows, 1, dim), torch.randn(rows, 1, dim)
    oc[1] *= 0.3           # a row whose quantile stays below 1 (s clamps to 1) ...
    oc[2] *= 3.0           # ... and one whose quantile exceeds sample_max_value
    vn = torch.randn(rows, 1, dim)
    a_t, a_prev = sch._alphas(t)
    p = hip.DenoiseParams(guidance_scale=3.0, guidance_rescale=0.7, ddim_eta=eta, epsilon_scaling=1.1, clip_noise=1.5,
                          init_noise_sigma=1.0, prediction_type=hip.PRED[ptype], clip_sample=0, clip_sample_range=1.0,
                          thresholding=int(thresholding), dynamic_thresholding_ratio=0.9, sample_max_value=1.7)
    kw = dict(eta=eta, epsilon_scaling=1.1, variance_noise=vn if eta > 0 else None)
    if mode == "cfg":
        class M:  # the oracle's guidance needs only these attributes
            guidance_scale, guidance_rescale = 3.0, 0.7
        eps = sch.get_epsilon(torch.cat([oc, ou]), torch.cat([x, x]), t)
        eps = oracle.OracleTwoTower.apply_cfg(M, eps)
        prev, x0 = sch.step(eps, t, x, prediction_type="epsilon", **kw)
    else:
        prev, x0 = sch.step(oc, t, x, prediction_type=ptype, **kw)
    prev = prev.clip(-1.5, 1.5)
    xg, ocg, oug = x.reshape(rows, dim).cuda(), oc.reshape(rows, dim).cuda(), ou.reshape(rows, dim).cuda()
    vng = vn.reshape(rows, dim).cuda() if eta > 0 else None
    pg, zg = torch.empty_like(xg), torch.empty_like(xg)
    hip.check(hip.lib.lcmhip_ddim_cfg_step(hip.ptr(xg), hip.ptr(ocg), hip.ptr(oug) if mode == "cfg" else None, hip.ptr(pg),
                                           hip.ptr(zg), rows, dim, float(a_t), float(a_prev), C.byref(p), hip.ptr(vng), _st(hip)))
    assert (pg.cpu() - prev.reshape(rows, dim)).abs().max() <= 2e-5
    assert (zg.cpu() - x0.reshape(rows, dim)).abs().max() <= 2e-5 * max(1.0, float(x0.abs().max()))
    if eta > 0:     # missing variance noise is an error, not a silent deterministic step
        assert hip.lib.lcmhip_ddim_cfg_step(hip.ptr(xg), hip.ptr(ocg), None, hip.ptr(pg), hip.ptr(zg), rows, dim, float(a_t),
                                            float(a_prev), C.byref(p), None, _st(hip)) == -1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("causal,s,base", [(0, 1, 1), (0, 1, 137), (1, 1, 40), (1, 7, 0), (1, 5, 66),
                                           (1, 16, 5), (1, 64, 0), (1, 130, 37), (1, 200, 0)])
def test_attention(hip, dtype, causal, s, base):
    from oracle.model import sdpa
    torch.manual_seed(5)
    n_docs, H, cap = 3, 8, 256
    d = H * 128
    kc = torch.randn(4, H, cap, 128).cuda().to(dtype)  # [slot][head][position][128]; documents -> slots 2, 0, 3
    vc = torch.randn(4, H, cap, 128).cuda().to(dtype)
    slot = torch.tensor([2, 0, 3], dtype=torch.int32).cuda()
    q = (2 * torch.randn(n_docs * s, d)).cuda().to(dtype)
    out = torch.empty_like(q)
    hip.check(hip.lib.lcmhip_attention(hip.ptr(q), hip.ptr(kc), hip.ptr(vc), hip.ptr(slot), hip.ptr(out), n_docs, s, H, cap,
                                       base, causal, hip.dtype_code(dtype), _st(hip)))
    nk = base + (s if causal else 0)
    qh = q.float().cpu().view(n_docs, s, H, 128).transpose(1, 2)
    kh = kc.float().cpu()[[2, 0, 3], :, :nk]
    vh = vc.float().cpu()[[2, 0, 3], :, :nk]
    mask = None
    if causal and s > 1:
        mask = torch.full((nk, nk), float("-inf")).triu(1)[nk - s:]
    ref = sdpa(qh, kh, vh, None, mask).transpose(1, 2).reshape(n_docs * s, d)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (out.float().cpu() - ref).abs().max() <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("causal,s,base", [(0, 1, 3801), (1, 1, 3800), (1, 96, 3705), (1, 3801, 0)])
def test_attention_at_the_context_limit(hip, dtype, causal, s, base):
    """The longest cache a session may hold: 3800 context positions + the zero slot = 3801 keys per document.  One
    query over all of them (the sampling loop's cross-attention at the limit), the last positions of a long prompt, and a
    whole 3801-position prompt through the prefill kernel (bf16) / the one-query kernel (fp32)."""
    from oracle.model import sdpa
    torch.manual_seed(11)
    n_docs, H, cap = 2, 4, 3801
    d = H * 128
    kc = torch.randn(3, H, cap, 128).cuda().to(dtype)
    vc = torch.randn(3, H, cap, 128).cuda().to(dtype)
    slot = torch.tensor([2, 0], dtype=torch.int32).cuda()
    q = (2 * torch.randn(n_docs * s, d)).cuda().to(dtype)
    out = torch.empty_like(q)
    hip.check(hip.lib.lcmhip_attention(hip.ptr(q), hip.ptr(kc), hip.ptr(vc), hip.ptr(slot), hip.ptr(out), n_docs, s, H, cap,
                                       base, causal, hip.dtype_code(dtype), _st(hip)))
    nk = base + (s if causal else 0)
    assert nk == cap
    qh = q.float().cpu().view(n_docs, s, H, 128).transpose(1, 2)
    kh = kc.float().cpu()[[2, 0]]
    vh = vc.float().cpu()[[2, 0]]
    mask = None
    if causal and s > 1:
        mask = torch.full((nk, nk), float("-inf")).triu(1)[nk - s:]
    ref = sdpa(qh, kh, vh, None, mask).transpose(1, 2).reshape(n_docs * s, d)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (out.float().cpu() - ref).abs().max() <= tol


@pytest.mark.parametrize("n_docs,H,s,base", [(3, 4, 30, 0), (5, 4, 70, 3), (1, 12, 129, 64), (7, 16, 64, 63)])
def test_attention_prefill_grid_shapes(hip, n_docs, H, s, base):
    """The matrix-core prefill kernel over grids that are and are not a multiple of the 8 XCDs (its tile order is
    remapped only in the first case), key ranges that end inside, at and just past a 64-key stage, one to three query
    tiles per (document, head)."""
    from oracle.model import sdpa
    torch.manual_seed(n_docs + H + s)
    dtype, cap = torch.bfloat16, 256
    d = H * 128
    kc = torch.randn(n_docs + 1, H, cap, 128).cuda().to(dtype)
    vc = torch.randn(n_docs + 1, H, cap, 128).cuda().to(dtype)
    slots = list(range(n_docs, 0, -1))
    slot = torch.tensor(slots, dtype=torch.int32).cuda()
    q = (2 * torch.randn(n_docs * s, d)).cuda().to(dtype)
    out = torch.empty_like(q)
    hip.check(hip.lib.lcmhip_attention(hip.ptr(q), hip.ptr(kc), hip.ptr(vc), hip.ptr(slot), hip.ptr(out), n_docs, s, H, cap,
                                       base, 1, hip.dtype_code(dtype), _st(hip)))
    nk = base + s
    qh = q.float().cpu().view(n_docs, s, H, 128).transpose(1, 2)
    kh = kc.float().cpu()[slots, :, :nk]
    vh = vc.float().cpu()[slots, :, :nk]
    mask = torch.full((nk, nk), float("-inf")).triu(1)[nk - s:]
    ref = sdpa(qh, kh, vh, None, mask).transpose(1, 2).reshape(n_docs * s, d)
    assert (out.float().cpu() - ref).abs().max() <= 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["plain", "extra_const", "dup_init", "bare"])
@pytest.mark.parametrize("kind", ["rms", "standard"])
@pytest.mark.parametrize("N", [768, 1024, 2048, 4096])
def test_gemm_partial_residual_norm(hip, dtype, mode, kind, N):
    """Split-K slab -> fused residual update + norm + modulation.  N = 1024 / 2048 / 4096 take the kernel that requests
    every operand in one round trip (absent operands masked), N = 768 the general one; "bare" = no bias, gate or
    modulation; "standard" = LayerNorm with weight and bias."""
    torch.manual_seed(11)
    M, K = 64, 2048
    rows_total = M if mode in ("plain", "bare") else 2 * M
    bare = mode == "bare"
    A = torch.randn(M, K).cuda().to(dtype)
    W = (torch.randn(N, K) / math.sqrt(K)).cuda().to(dtype)
    bias = torch.randn(N).cuda().to(dtype)
    gate, const = torch.randn(N).cuda(), torch.randn(N).cuda()
    w = (1 + 0.1 * torch.randn(N)).cuda().to(dtype)
    nb = (0.2 * torch.randn(N)).cuda().to(dtype)
    shift, scale = (0.3 * torch.randn(N)).cuda(), (0.3 * torch.randn(N)).cuda()
    x0 = torch.randn(rows_total, N).cuda()
    slab = torch.empty(12 * M * N, device="cuda")
    splits = C.c_int32(0)
    hip.check(hip.lib.lcmhip_gemm_partial(hip.ptr(A), K, hip.ptr(W), hip.ptr(slab), slab.numel() * 4, C.byref(splits),
                                          M, N, K, hip.dtype_code(dtype), _st(hip)))
    assert splits.value >= 2
    lin = (A.double() @ W.double().T + (0 if bare else bias.double())).cpu()
    x = x0.clone()
    y = torch.empty(rows_total, N, device="cuda", dtype=dtype)
    rows_norm = M if mode == "extra_const" else rows_total
    layernorm = kind == "standard"
    hip.check(hip.lib.lcmhip_residual_norm(
        hip.ptr(x), hip.ptr(slab), splits.value, M, None if bare else hip.ptr(bias), None if bare else hip.ptr(gate),
        hip.ptr(const) if mode == "extra_const" else None, int(mode == "dup_init"), int(mode == "dup_init"),
        hip.ptr(w), hip.ptr(nb) if layernorm else None, None if bare else hip.ptr(shift), None if bare else hip.ptr(scale),
        hip.ptr(y), rows_total, rows_norm, N, hip.NORM_LAYERNORM if layernorm else hip.NORM_RMS, 1e-5,
        hip.dtype_code(dtype), _st(hip)))
    g = torch.ones(N, dtype=torch.float64) if bare else gate.double().cpu()
    if mode in ("plain", "bare"):
        ref_x = x0.double().cpu() + g * lin
    elif mode == "extra_const":
        ref_x = x0.double().cpu() + torch.cat([g * lin, (g * const.double().cpu()).expand(M, N)])
    else:
        ref_x = torch.cat([g * lin, g * lin])
    tol = 2e-4 if dtype == torch.float32 else 3e-3
    assert (x.double().cpu() - ref_x).abs().max() <= tol * max(1.0, float(ref_x.abs().max()))
    xr = ref_x[:rows_norm]
    if layernorm:
        ref_y = torch.nn.functional.layer_norm(xr, (N,), w.double().cpu(), nb.double().cpu(), 1e-5)
    else:
        ref_y = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5) * w.double().cpu()
    if not bare:
        ref_y = ref_y * (1 + scale.double().cpu()) + shift.double().cpu()
    tol_y = 1e-4 if dtype == torch.float32 else 4e-2
    assert (y[:rows_norm].double().cpu() - ref_y).abs().max() <= tol_y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_finalize_step(hip, dtype):
    """The tensor work of the generators' finalize_step against plain torch: prompt rows keep their embedding, other
    rows are stored in the buffer's dtype, both stopping-rule cosines are torch.nn.functional.cosine_similarity."""
    torch.manual_seed(11)
    n_docs, total, dim, step = 7, 6, 1024, 3
    seqs = (torch.randn(n_docs, total, dim, device="cuda") * 0.02).to(dtype)
    before = seqs.clone()
    docs = torch.tensor([0, 2, 3, 6], device="cuda", dtype=torch.int32)
    plens = torch.tensor([2, 9, 5, 3, 9, 9, 4], device="cuda", dtype=torch.int32)        # documents 2 and 6 still in their prompt
    new = torch.randn(4, dim, device="cuda") * 0.02
    new[1] = before[2, step - 1].float() * 1.5                                             # a repetition (cosine 1) ...
    eos = new[0].clone() * 3.0                                                             # ... and an EOS hit for row 0
    cos = torch.full((4, 2), -7.0, device="cuda")
    hip.check(hip.lib.lcmhip_finalize_step(hip.ptr(new), hip.ptr(seqs), hip.dtype_code(dtype), hip.ptr(docs), hip.ptr(plens),
                                           4, step, total, dim, hip.ptr(eos), hip.ptr(cos), _st(hip)))
    torch.cuda.synchronize()
    kept = torch.stack([before[2, step].float() if d in (2, 6) and False else new[r] for r, d in enumerate(docs.tolist())])
    kept[1], kept[3] = before[2, step].float(), before[6, step].float()                    # in-prompt rows keep the prompt
    want = before.clone()
    for r, d in enumerate(docs.tolist()):
        want[d, step] = kept[r].to(dtype)
    assert torch.equal(seqs, want)
    prev = torch.stack([before[d, step - 1].float() for d in docs.tolist()])
    c_eos = torch.nn.functional.cosine_similarity(eos, kept)
    c_prev = torch.nn.functional.cosine_similarity(prev, kept)
    assert (cos[:, 0] - c_eos).abs().max() <= 1e-5 and (cos[:, 1] - c_prev).abs().max() <= 1e-5
    assert cos[0, 0] > 0.9999
    # no row map, no prompt lengths, no EOS vector, first step: cosines are reported as 0
    cos.fill_(-7.0)
    hip.check(hip.lib.lcmhip_finalize_step(hip.ptr(new), hip.ptr(seqs), hip.dtype_code(dtype), None, None, 4, 0, total, dim,
                                           None, hip.ptr(cos), _st(hip)))
    torch.cuda.synchronize()
    assert torch.equal(seqs[:4, 0], new.to(dtype)) and float(cos.abs().max()) == 0.0
    assert hip.lib.lcmhip_finalize_step(hip.ptr(new), hip.ptr(seqs), hip.dtype_code(dtype), None, None, 4, total, total, dim,
                                        None, hip.ptr(cos), _st(hip)) == -1


@pytest.mark.parametrize("dim", [1024, 2048])
@pytest.mark.parametrize("mode", ["plain_sample", "cfg"])
def test_ddim_cfg_step_many_rows(hip, mode, dim):
    """From 1024 rows of finished outputs the update runs one wave per row; same arithmetic as the one-workgroup-per-row
    kernel (checked against it and against the oracle's scheduler), with and without the predicted-x0 output."""
    import oracle
    from oracle.archs import SchedulerConfig
    torch.manual_seed(5)
    rows = 1027
    sch = oracle.DDIMSchedule(SchedulerConfig(num_train_steps=100))
    sch.set_timesteps(40)
    t = 31
    x = torch.randn(rows, 1, dim) * 0.8
    oc, ou, vn = torch.randn(rows, 1, dim), torch.randn(rows, 1, dim), torch.randn(rows, 1, dim)
    a_t, a_prev = sch._alphas(t)
    p = hip.DenoiseParams(guidance_scale=3.0, guidance_rescale=0.7, ddim_eta=0.3, epsilon_scaling=0.0, clip_noise=2.5,
                          init_noise_sigma=1.0, prediction_type=hip.PRED["sample"], clip_sample=0, clip_sample_range=1.0,
                          thresholding=0, dynamic_thresholding_ratio=0.995, sample_max_value=1.0)
    if mode == "cfg":
        class M:
            guidance_scale, guidance_rescale = 3.0, 0.7
        eps = oracle.OracleTwoTower.apply_cfg(M, sch.get_epsilon(torch.cat([oc, ou]), torch.cat([x, x]), t))
        prev, x0 = sch.step(eps, t, x, prediction_type="epsilon", eta=0.3, variance_noise=vn)
    else:
        prev, x0 = sch.step(oc, t, x, eta=0.3, variance_noise=vn)
    prev = prev.clip(-2.5, 2.5)
    g = lambda a: a.reshape(rows, dim).cuda()        # noqa: E731
    xg, ocg, oug, vng = g(x), g(oc), g(ou), g(vn)
    pg, zg = torch.empty_like(xg), torch.empty_like(xg)
    hip.check(hip.lib.lcmhip_ddim_cfg_step(hip.ptr(xg), hip.ptr(ocg), hip.ptr(oug) if mode == "cfg" else None, hip.ptr(pg),
                                           hip.ptr(zg), rows, dim, float(a_t), float(a_prev), C.byref(p), hip.ptr(vng), _st(hip)))
    assert (pg.cpu() - prev.reshape(rows, dim)).abs().max() <= 2e-5
    assert (zg.cpu() - x0.reshape(rows, dim)).abs().max() <= 2e-5 * max(1.0, float(x0.abs().max()))
    p2 = torch.empty_like(xg)
    hip.check(hip.lib.lcmhip_ddim_cfg_step(hip.ptr(xg), hip.ptr(ocg), hip.ptr(oug) if mode == "cfg" else None, hip.ptr(p2),
                                           None, rows, dim, float(a_t), float(a_prev), C.byref(p), hip.ptr(vng), _st(hip)))
    assert torch.equal(p2, pg)
    # the first 5 rows through the small-batch kernel
    p3, z3 = torch.empty(5, dim, device="cuda"), torch.empty(5, dim, device="cuda")
    hip.check(hip.lib.lcmhip_ddim_cfg_step(hip.ptr(xg), hip.ptr(ocg), hip.ptr(oug) if mode == "cfg" else None, hip.ptr(p3),
                                           hip.ptr(z3), 5, dim, float(a_t), float(a_prev), C.byref(p), hip.ptr(vng), _st(hip)))
    assert (p3 - pg[:5]).abs().max() <= 1e-5


_ACT_MODULES = {"relu": torch.nn.ReLU(), "tanh": torch.nn.Tanh(), "elu": torch.nn.ELU(), "leaky_relu": torch.nn.LeakyReLU(),
                "selu": torch.nn.SELU(), "gelu": torch.nn.GELU(), "silu": torch.nn.SiLU(), "softsign": torch.nn.Softsign(),
                "sigmoid": torch.nn.Sigmoid(), "hardsigmoid": torch.nn.Hardsigmoid()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("act", sorted(_ACT_MODULES))
def test_gemm_activation_epilogues(hip, dtype, act):
    """out = act(A W^T + b) for every inner activation the reference's two-matrix FFN can name
    (lcm/nn/initialization.py:121-151), against the torch module of that name on a float64 product.  Shapes take the
    call through the weight-streaming tiles (50 rows, with and without a K split + finishing pass) and, in bf16, the
    many-row matrix-core kernels (2048 and 4096 rows)."""
    code = hip.EPI_RELU | (hip.ACT[act] << 8)
    shapes = [(50, 1024, 512, False), (50, 1024, 2048, True)]
    if dtype == torch.bfloat16:
        shapes += [(2048, 2048, 512, False), (4096, 4096, 256, False)]
    for M, N, K, split in shapes:
        torch.manual_seed(M + N + K)
        A = torch.randn(M, K).cuda().to(dtype)
        W = (1.5 * torch.randn(N, K) / math.sqrt(K)).cuda().to(dtype)       # pre-activations of both signs, |v| up to ~5
        b = torch.randn(N).cuda().to(dtype)
        out = torch.empty(M, N, device="cuda", dtype=dtype)
        ws = torch.empty(16 * M * N, device="cuda") if split else None
        _gemm(hip, A, W, None, b, None, None, out, code, ws)
        torch.cuda.synchronize()
        ref = _ACT_MODULES[act](A.double().cpu() @ W.double().cpu().T + b.double().cpu())
        tol = 2e-5 if dtype == torch.float32 else 1.2e-2
        err = (out.double().cpu() - ref).abs().max()
        assert err <= tol * max(1.0, float(ref.abs().max())), (act, M, N, K, float(err))
    # an activation code on any other epilogue is refused
    rc = hip.lib.lcmhip_gemm(hip.ptr(A), K, hip.ptr(W), None, None, None, None, hip.ptr(out), N, M, N, K,
                             hip.EPI_STORE | (hip.ACT["gelu"] << 8), hip.dtype_code(dtype), None, 0, _st(hip))
    assert rc == -1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_stand_alone_timestep_encoder(dtype):
    """`lcm.nn.timestep_encoder.DiTTimestepEncoder` (lcm/nn/timestep_encoder.py:18-122) used by itself: sinusoidal
    frequencies, Linear, SiLU, Linear through `lcmhip_gemm`, against the oracle's `embed_time` on the same weights."""
    import oracle as O
    from lcm.nn.timestep_encoder import DiTTimestepEncoder
    cfg = O.get_arch("toy_two_tower_diffusion_lcm")
    sd = O.init_state_dict(cfg, seed=3, t_gain=3.0)
    enc = DiTTimestepEncoder(1024, device="cuda:0", dtype=dtype)
    assert set(enc.state_dict()) == {"fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"}
    assert float(enc.state_dict()["fc1.bias"].abs().max()) == 0.0 and 0.015 < float(enc.state_dict()["fc2.weight"].float().std()) < 0.025
    enc.load_state_dict({k[len("denoiser.embed_time."):]: v for k, v in sd.items() if k.startswith("denoiser.embed_time.")})
    ts = torch.tensor([[0, 1, 17], [250, 999, 500]])
    if dtype == torch.bfloat16:
        sd = {k: v.bfloat16().float() for k, v in sd.items()}
    ref = O.OracleTwoTower(cfg, sd).embed_time(ts)
    out = enc(ts.cuda())
    assert out.shape == (2, 3, 1024) and out.dtype == dtype
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (out.float().cpu() - ref).abs().max() <= tol * max(1.0, float(ref.abs().max()))
    emb = DiTTimestepEncoder.sinusoidal_timestep_embedding(torch.tensor([3.0, 7.5]), 7)
    assert emb.shape == (2, 7) and float(emb[:, -1].abs().max()) == 0.0 and abs(float(emb[0, 0]) - math.cos(3.0)) < 1e-6
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        DiTTimestepEncoder(1024, device="cpu")


@pytest.mark.parametrize("pred", ["sample", "v_prediction", "epsilon"])
def test_scheduler_get_epsilon(pred):
    """`DDIMScheduler.get_epsilon` of the product (ddim.py:561-587) against the oracle's scheduler, at an ordinary
    timestep and at the last training step, where the zero-terminal-SNR schedule has alpha_bar = 0."""
    import oracle as O
    from lcm.nn.schedulers import DDIMScheduler, DDIMSchedulerConfig
    from oracle.archs import SchedulerConfig
    a = DDIMScheduler(DDIMSchedulerConfig(num_diffusion_train_steps=100, prediction_type=pred))
    b = O.DDIMSchedule(SchedulerConfig(num_train_steps=100, prediction_type=pred))
    g = torch.Generator().manual_seed(3)
    x, out = torch.randn(5, 2, 1024, generator=g), torch.randn(5, 2, 1024, generator=g)
    for t in (37, 99):
        want = b.get_epsilon(out, x, t)
        got = a.get_epsilon(out.cuda(), x.cuda(), t).cpu()
        assert got.shape == want.shape and torch.isfinite(got).all()
        assert (got - want).abs().max() <= 1e-5 * max(1.0, float(want.abs().max())), (pred, t)


def test_scheduler_add_noise_and_velocity():
    """`DDIMScheduler.add_noise` / `get_velocity` (reference ddim.py:512-559) through the step kernel, one timestep per
    leading entry of the samples, against the closed forms in fp32 -- flat rows, documents of several positions sharing
    a timestep, a timestep per position, repeated and boundary timesteps (alpha_bar = 0 at the last step of the
    zero-terminal-SNR schedule)."""
    from lcm.nn.schedulers import DDIMScheduler, DDIMSchedulerConfig
    sch = DDIMScheduler(DDIMSchedulerConfig(num_diffusion_train_steps=100))
    ac = sch.alphas_cumprod.float().cpu()
    g = torch.Generator().manual_seed(12)
    cases = [((9, 1024), torch.tensor([5, 99, 0, 5, 42, 42, 7, 99, 63])),
             ((4, 3, 1024), torch.tensor([17, 0, 99, 17])),
             ((2, 5, 1024), torch.randint(0, 100, (2, 5), generator=g)),
             ((1, 1024), torch.tensor([50]))]
    for shape, t in cases:
        x, nz = torch.randn(shape, generator=g) * 3.0, torch.randn(shape, generator=g)
        a = ac[t.flatten()].repeat_interleave((x.numel() // 1024) // t.numel()).reshape(shape[:-1] + (1,))
        want_noisy = a.sqrt() * x + (1 - a).sqrt() * nz
        want_vel = a.sqrt() * nz - (1 - a).sqrt() * x
        got_noisy = sch.add_noise(x.cuda(), nz.cuda(), t).cpu()
        got_vel = sch.get_velocity(x.cuda(), nz.cuda(), t.cuda()).cpu()
        assert got_noisy.shape == x.shape and got_vel.shape == x.shape
        assert (got_noisy - want_noisy).abs().max() <= 2e-6 * float(want_noisy.abs().max()), shape
        assert (got_vel - want_vel).abs().max() <= 2e-6 * float(want_vel.abs().max()), shape
    with pytest.raises(RuntimeError):
        sch.add_noise(torch.zeros(2, 1024), torch.zeros(2, 1024), torch.tensor([1, 2]))      # host tensors: no CPU path
