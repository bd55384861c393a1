"""Several host threads generating at once from one model (one stream, session and generator per thread): every thread
must get the bits the same call gives when it runs alone.  Collected near the end of the suite on purpose: a failure
of host-side thread safety can take the interpreter down, and `pytest -x` would then hide every test behind it."""

import pytest
import torch

import oracle as oracle_mod
from parity_utils import DEFAULT_OPTS, make_problem, run_hip

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
