"""Several host threads generating at once from one model (one stream, session and generator per thread): every thread
must get the bits the same call gives when it runs alone.  Collected near the end of the suite on purpose: a failure
of host-side thread safety can take the interpreter down, and `pytest -x` would then hide every test behind it."""

import pytest
import torch

import oracle as oracle_mod
from parity_utils import DEFAULT_OPTS, make_base_problem, make_problem, run_hip, run_hip_base

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_generators_on_concurrent_host_threads(dtype):
    """A serving process: four host threads share ONE model; each has its own HIP stream, generator and batch size,
    so sessions are created, recorded into graphs, pooled and destroyed by one thread while the others record or
    replay theirs.  Every thread must get, three times over, the bits the same call gives when it runs alone."""
    import threading
    arch = "mini_two_tower_diffusion_lcm"
    sizes = [3, 2, 4, 1]
    cfg, sd, x, nz = make_problem(arch, sum(sizes), 3, 5)
    if dtype == torch.bfloat16:
        sd = {k: oracle_mod.r16(v) for k, v in sd.items()}
        x = oracle_mod.r16(x)
    opts = dict(DEFAULT_OPTS, inference_timesteps=10)
    bounds = [(sum(sizes[:i]), sum(sizes[:i + 1])) for i in range(len(sizes))]
    model, alone = None, []
    for lo, hi in bounds:
        out, model = run_hip(arch, sd, x[lo:hi], nz[:, lo:hi], 5, opts, dtype=dtype, model=model)
        alone.append(torch.stack([h[0].seq for h in out.hypotheses]).cpu())
    errors, streams = [], [torch.cuda.Stream("cuda:0") for _ in sizes]
    start = threading.Barrier(len(sizes))

    def work(i):
        try:
            lo, hi = bounds[i]
            torch.cuda.set_device(0)
            start.wait()
            with torch.cuda.stream(streams[i]):
                for rep in range(3):
                    out, _ = run_hip(arch, sd, x[lo:hi], nz[:, lo:hi], 5, opts, dtype=dtype, model=model)
                    got = torch.stack([h[0].seq for h in out.hypotheses]).cpu()
                    if not torch.equal(got, alone[i]):
                        errors.append((i, rep, float((got.float() - alone[i].float()).abs().max())))
        except Exception as e:  # noqa: BLE001  (reported below, in the main thread)
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(sizes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    del model


@pytest.mark.parametrize("disable_cache", [False, True])
def test_base_generators_on_concurrent_host_threads(disable_cache):
    """The same for the base LCM's generator (one model, four threads, streams and batch sizes), with its KV cache and
    with the full re-encode per step, which takes a throw-away session from the model's pool at every step."""
    import threading
    arch = "mini_base_lcm"
    sizes = [3, 2, 4, 1]
    cfg, sd, x, nz = make_base_problem(arch, sum(sizes), 3, 4)
    opts = dict(eos_threshold=None, lcm_temperature=0.5, sample_latent_variable=True)
    bounds = [(sum(sizes[:i]), sum(sizes[:i + 1])) for i in range(len(sizes))]
    model, alone = None, []
    for lo, hi in bounds:
        out, model = run_hip_base(arch, sd, x[lo:hi], nz[:, lo:hi], 4, opts, model=model, disable_cache=disable_cache)
        alone.append(torch.stack([h[0].seq for h in out.hypotheses]).cpu())
    errors, streams = [], [torch.cuda.Stream("cuda:0") for _ in sizes]
    start = threading.Barrier(len(sizes))

    def work(i):
        try:
            lo, hi = bounds[i]
            torch.cuda.set_device(0)
            start.wait()
            with torch.cuda.stream(streams[i]):
                for rep in range(3):
                    out, _ = run_hip_base(arch, sd, x[lo:hi], nz[:, lo:hi], 4, opts, model=model, disable_cache=disable_cache)
                    got = torch.stack([h[0].seq for h in out.hypotheses]).cpu()
                    if not torch.equal(got, alone[i]):
                        errors.append((i, rep, float((got.float() - alone[i].float()).abs().max())))
        except Exception as e:  # noqa: BLE001  (reported below, in the main thread)
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(sizes))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    del model
