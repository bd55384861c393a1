"""The C ABI from a process without Python or torch: tests/c_caller/abi_caller.c (plain C99 built by gcc against
include/lcmhip.h, the HIP runtime its only other dependency) runs prefill -> sampling loop -> append -> sampling loop
on a model it builds from a case file, and must return the bits the Python host gets from the same calls."""

import ctypes as C
import os
import struct
import subprocess

import pytest
import torch

from parity_utils import DEFAULT_OPTS, base_config, mini_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLER = os.path.join(ROOT, "tests", "c_caller", "abi_caller")

pytestmark = pytest.mark.gpu


def write_case(path, model, session_params, timesteps, alphas, final_alpha, step_ratio, prompts, noise, max_ctx):
    with open(path, "wb") as f:
        desc = model.model_desc()
        f.write(struct.pack("<i", 0x4C434D31))
        f.write(struct.pack("<i", C.sizeof(desc)) + bytes(desc))
        f.write(struct.pack("<i", C.sizeof(session_params)) + bytes(session_params))
        f.write(struct.pack(f"<i{len(timesteps)}i", len(timesteps), *timesteps))
        f.write(struct.pack(f"<i{len(alphas)}f", len(alphas), *alphas))
        f.write(struct.pack("<fi", final_alpha, step_ratio))
        f.write(struct.pack("<iii", prompts.shape[0], prompts.shape[1], max_ctx))
        f.write(prompts.cpu().float().contiguous().numpy().tobytes())
        f.write(noise.cpu().float().contiguous().numpy().tobytes())
        tensors = model.bound_tensors()
        f.write(struct.pack("<i", len(tensors)))
        for name, t in tensors.items():
            raw = t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes() if t.numel() else b""
            f.write(struct.pack("<i", len(name)) + name.encode())
            f.write(struct.pack(f"<i{t.dim()}q", t.dim(), *t.shape))
            f.write(struct.pack("<q", len(raw)) + raw)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_c_caller_without_python_gets_the_same_bits(tmp_path, dtype):
    from lcm import _hip
    from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model
    from lcm.models.two_tower_diffusion_lcm.builder import DeviceSession

    if not os.path.exists(CALLER):          # normally built by __graft_entry__.build(); one gcc call
        subprocess.run(["make", "-C", os.path.dirname(CALLER), "abi_caller"], check=True, capture_output=True)
    dev = torch.device("cuda", 0)
    model = create_two_tower_diffusion_lcm_model(mini_config(), device=dev, dtype=dtype, seed=0)
    model.init_parameters(seed=3, modulator_std=0.2)
    model.prep_for_denoising(DiffusionLCMGeneratorOptions(**DEFAULT_OPTS))
    sch = model.noise_scheduler
    n_docs, prompt_len, max_ctx = 5, 7, 16
    g = torch.Generator().manual_seed(11)
    prompts = (torch.randn(n_docs, prompt_len, 1024, generator=g) * 0.006).to(dev)
    noise = torch.randn(2, n_docs, 1024, generator=g).to(dev)

    # the Python host: the same five calls through ctypes
    n_steps = len(sch.timesteps)
    sess = DeviceSession(model, n_docs, max_ctx, n_steps)
    sess.set_schedule()
    sess.reset(n_docs)
    sess.append(prompts, normalized=False)
    first = sess.denoise(noise[0])
    sess.append(first.unsqueeze(1), normalized=False)
    second = sess.denoise(noise[1])
    torch.cuda.current_stream().synchronize()
    want = torch.stack([first, second]).cpu()
    assert torch.isfinite(want).all() and not torch.equal(want[0], want[1])

    params = _hip.DenoiseParams(
        guidance_scale=float(model.guidance_scale), guidance_rescale=float(model.guidance_rescale),
        ddim_eta=float(model.ddim_eta), epsilon_scaling=float(model.epsilon_scaling) if model.epsilon_scaling else 0.0,
        clip_noise=float(model.clip_noise), init_noise_sigma=float(sch.init_noise_sigma),
        prediction_type=_hip.PRED[sch.prediction_type], clip_sample=int(sch.config.clip_sample),
        clip_sample_range=float(sch.config.clip_sample_range), thresholding=int(sch.config.thresholding),
        dynamic_thresholding_ratio=float(sch.config.dynamic_thresholding_ratio),
        sample_max_value=float(sch.config.sample_max_value))
    case, out = tmp_path / "case.bin", tmp_path / "out.bin"
    write_case(case, model, params, [int(t) for t in sch.timesteps.tolist()], sch.alphas_cumprod.tolist(),
               float(sch.final_alpha_cumprod), int(sch.step_ratio), prompts, noise, max_ctx)
    keep = os.environ.get("LCM_CASE_DIR")       # for runs of other builds of the caller (sanitizer builds) on the same case
    if keep:
        import shutil
        tag = "bf16" if dtype == torch.bfloat16 else "fp32"
        shutil.copy(case, os.path.join(keep, f"case_{tag}.bin"))
        with open(os.path.join(keep, f"want_{tag}.bin"), "wb") as f:
            f.write(want.contiguous().numpy().tobytes())
    run = subprocess.run([CALLER, str(case), str(out)], capture_output=True, text=True, timeout=240)
    assert run.returncode == 0, (run.returncode, run.stdout[-2000:], run.stderr[-2000:])
    assert run.stdout.startswith("ok 5 docs, context 8"), run.stdout
    got = torch.frombuffer(bytearray(out.read_bytes()), dtype=torch.float32).reshape(2, n_docs, 1024)
    assert torch.equal(got, want), float((got - want).abs().max())

    # four host threads of the C process at once, each with a session and a stream of its own on the one model, three
    # sessions in a row per thread: the caller itself requires equal bits from every thread
    out4 = tmp_path / "out4.bin"
    run = subprocess.run([CALLER, str(case), str(out4), "4"], capture_output=True, text=True, timeout=240)
    assert run.returncode == 0, (run.returncode, run.stdout[-2000:], run.stderr[-2000:])
    assert run.stdout.startswith("ok 5 docs, context 8") and "4 threads" in run.stdout, run.stdout
    got = torch.frombuffer(bytearray(out4.read_bytes()), dtype=torch.float32).reshape(2, n_docs, 1024)
    assert torch.equal(got, want), float((got - want).abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_c_caller_base_lcm(tmp_path, dtype):
    """The base LCM through `lcmhip_base_forward` from the C process: the sampled prediction after a prompt, then after
    that sentence as one more position -- the bits of the Python host, from one thread and from four."""
    from lcm import _hip
    from lcm.models.base_lcm import create_base_lcm_model
    from lcm.models.two_tower_diffusion_lcm.builder import DeviceSession

    if not os.path.exists(CALLER):
        subprocess.run(["make", "-C", os.path.dirname(CALLER), "abi_caller"], check=True, capture_output=True)
    dev = torch.device("cuda", 0)
    model = create_base_lcm_model(base_config("mini_base_lcm"), device=dev, dtype=dtype, seed=2)
    n_docs, prompt_len, max_ctx = 5, 7, 16
    g = torch.Generator().manual_seed(13)
    prompts = (torch.randn(n_docs, prompt_len, 1024, generator=g) * 0.006).to(dev)
    noise = torch.randn(2, n_docs, 1024, generator=g).to(dev)

    sess = DeviceSession(model, n_docs, max_ctx, 1)
    sess.reset(n_docs)
    first = sess.base_forward(prompts, normalized=False, noise=noise[0], temperature=0.5, all_positions=False)
    second = sess.base_forward(first.unsqueeze(1), normalized=False, noise=noise[1], temperature=0.5, all_positions=False)
    torch.cuda.current_stream().synchronize()
    want = torch.stack([first, second]).cpu()
    assert torch.isfinite(want).all() and not torch.equal(want[0], want[1])

    case = tmp_path / "case.bin"
    write_case(case, model, _hip.DenoiseParams(), [0], [1.0], 1.0, 1, prompts, noise, max_ctx)
    for threads in ("1", "4"):
        out = tmp_path / f"out{threads}.bin"
        run = subprocess.run([CALLER, str(case), str(out), threads], capture_output=True, text=True, timeout=240)
        assert run.returncode == 0, (run.returncode, run.stdout[-2000:], run.stderr[-2000:])
        assert run.stdout.startswith("ok 5 docs, context 8, 1 steps"), run.stdout
        got = torch.frombuffer(bytearray(out.read_bytes()), dtype=torch.float32).reshape(2, n_docs, 1024)
        assert torch.equal(got, want), (threads, float((got - want).abs().max()))
    keep = os.environ.get("LCM_CASE_DIR")
    if keep:
        import shutil
        tag = "base_bf16" if dtype == torch.bfloat16 else "base_fp32"
        shutil.copy(case, os.path.join(keep, f"case_{tag}.bin"))
        with open(os.path.join(keep, f"want_{tag}.bin"), "wb") as f:
            f.write(want.contiguous().numpy().tobytes())
