#!/usr/bin/env python
"""Headline benchmark: sentence-embeddings generated/s, 1.6B two-tower LCM, 40 diffusion steps.

One "step" is one full `TwoTowerDiffusionLCMGenerator(batch)` call on the headline workload:
64 documents per GPU, 8 prompt sentences, 128 generated sentences, 40 requested DDIM
steps (50 executed, because the reference's integer `arange` truncates the step of the trailing
schedule), classifier-free guidance w=3 / rescale 0.7, bf16 weights, synthetic prompts, seeded
random-init weights and host-supplied noise indexed by global document id.  With N > 1 ranks
every rank generates its own 64 documents (weak scaling, no collective in the loop) and the
outputs are all-gathered once at the end (RCCL), inside the timed region.

Rank 0 prints ONE JSON line with the whole-job throughput plus
  roofline     -- the dominant kernel (weight-streaming GEMM of the denoiser FFN up-projection),
                  timed live with HIP events on the generation stream in an extra, untimed pass
                  (per-launch event pairs inside the loop for the kernel table, back-to-back launches
                  on the model's own weights for the kernel's average duration);
  cpu_baseline -- the CPU oracle (a restatement of the reference algorithm) timed on this box's
                  host cores on a bounded sample of the same workload.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak
ARCH = "two_tower_diffusion_lcm_1_6B"
OPTS = dict(guidance_scale=3.0, guidance_rescale=0.7, initial_noise_scale=0.6, inference_timesteps=40,
            eos_threshold=None, max_seq_len=128)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="documents per GPU")
    ap.add_argument("--prompt-len", type=int, default=8)
    ap.add_argument("--gen-len", type=int, default=128)
    ap.add_argument("--arch", default=ARCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sentence-steps", type=int, default=3,
                    help="generated sentences per document the CPU baseline runs after its prefill")
    ap.add_argument("--no-check", action="store_true", help="skip the cross-check of the timed output")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch the sampling loop kernel by kernel instead of replaying it from a HIP graph (profiler counter passes)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="ranks only rendezvous over gloo on the CPU and report the world size (no GPU work)")
    return ap.parse_args()


def visible_gpu_count():
    """GPUs this process tree may use, from the environment and the kernel driver's topology files only: the parent
    of the ranks must not load the GPU runtime.  None when it cannot be told."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip()])
    import glob
    n, seen = 0, False
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        seen = True
        n += int(props.get("simd_count", "0")) > 0          # CPU nodes have no SIMDs
    return n if seen else None


def launch_ranks(args):
    """`--gpus N` without a launcher around it: start N ranks as ONE child (`python -m torch.distributed.run`)
    and leave with its exit code.  This parent never calls into the GPU runtime (the device count comes from the
    environment or the driver's topology files), so nothing that has initialised the GPU is ever replaced or forked.  Under a launcher (WORLD_SIZE set) the world size must equal --gpus."""
    world_env = os.environ.get("WORLD_SIZE")
    if world_env is not None:
        if int(world_env) != args.gpus:
            print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks", file=sys.stderr)
            sys.exit(2)
        return
    if args.gpus == 1:
        return
    import socket
    import subprocess
    if not args.launcher_selftest and not os.environ.get("LCM_BENCH_ONE_DEVICE"):
        n_dev = visible_gpu_count()               # standard library only: nothing here touches the GPU runtime
        if n_dev is not None and n_dev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {n_dev} GPU(s) are visible", file=sys.stderr)
            sys.exit(2)
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def launcher_selftest(args):
    """What the launcher hands every rank, checked without a GPU: rendezvous over gloo, gather the ranks."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group("gloo")
        ranks = [None] * world
        dist.all_gather_object(ranks, dist.get_rank())
        n = dist.get_world_size()
    else:
        ranks, n = [0], 1
    assert n == args.gpus and sorted(ranks) == list(range(n)), (n, ranks)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": n, "ranks": sorted(ranks)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def make_noise(gen_len, batch, first_doc, device):
    """N(0,1) draws indexed by GLOBAL document id: independent of how documents are sharded."""
    out = torch.empty(gen_len, batch, 1024, device=device)
    for i in range(batch):
        g = torch.Generator(device=device).manual_seed(4321 + first_doc + i)
        out[:, i] = torch.randn(gen_len, 1024, generator=g, device=device)
    return out


def make_prompts(model, batch, prompt_len, first_doc, device):
    out = torch.empty(batch, prompt_len, 1024, device=device)
    c, s = model.sonar_normalizer.center.float(), model.sonar_normalizer.scale.float()
    for i in range(batch):
        g = torch.Generator(device=device).manual_seed(1234 + first_doc + i)
        out[i] = c + s * torch.randn(prompt_len, 1024, generator=g, device=device)
    return out


def roofline_pass(model, gen_cls, opt_cls, batch, prompt_len, gen_len, device):
    """Untimed extra pass with per-launch HIP events on the generation stream (mid-generation
    context length, two sentence steps).  Returns (roofline dict, per-kernel table)."""
    from lcm.datasets.batch import EmbeddingsBatch

    mid = prompt_len + gen_len // 2
    gen = gen_cls(model, opt_cls(**OPTS))
    x = make_prompts(model, batch, mid, 0, device).to(model.dtype)
    nz = make_noise(2, batch, 0, device)
    # the session is created lazily by the first call: run once plain, then once with events on
    gen(EmbeddingsBatch(x, None), max_gen_len=2, noise=nz)
    sess = model._session_pool[0]
    sess.profile_enable(True)
    gen(EmbeddingsBatch(x, None), max_gen_len=2, noise=nz)
    rows = sess.profile()
    sess.profile_enable(False)
    table = {}
    for r in rows:
        if r["launches"] == 0:
            continue
        avg_us = 1e3 * r["total_ms"] / r["launches"]
        table[r["name"]] = dict(launches=r["launches"], total_ms=round(r["total_ms"], 3), avg_us=round(avg_us, 3),
                                gbps=round(r["bytes"] / r["total_ms"] / 1e6, 1) if r["total_ms"] > 0 else None,
                                tflops=round(r["flops"] / r["total_ms"] / 1e9, 2) if r["total_ms"] > 0 else None)
    # dominant kernel of the denoising loop: the one that moves the largest share of the algorithmic HBM bytes
    # (den.ffn_in; also the largest share of time in an unprofiled run -- under rocprofv3 the per-launch
    # overhead of the many small norm kernels can exceed it, which must not change what is reported)
    dom = max((r for r in rows if r["name"].startswith("den.")), key=lambda r: r["bytes"])
    achieved = dom["bytes"] / dom["total_ms"] / 1e6      # GB/s
    roof = dict(bound="hbm", kernel=dom["name"], achieved=round(achieved, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                frac=round(achieved / HBM_PEAK_GBS, 4), traffic=None,
                avg_launch_us=round(1e3 * dom["total_ms"] / dom["launches"], 3),
                bytes_per_launch=round(dom["bytes"] / dom["launches"]))
    roof.update(pmc_traffic(dom["name"]))
    if dom["name"] == "den.ffn_in":
        # The event pair around a single launch inside the loop also spans about 2 us of launch boundary and event
        # processing; the kernel's own average duration is taken from back-to-back launches between one event pair
        # (it agrees with the rocprofv3 average), the in-loop figure is kept beside it.
        b2b = replay_ffn_in(model, 2 * batch, device)
        if b2b:
            roof["in_loop_event_pair_us"] = roof["avg_launch_us"]
            roof["avg_launch_us"] = b2b["back_to_back_us"]
            roof["achieved"] = round(roof["bytes_per_launch"] / b2b["back_to_back_us"] / 1e3, 1)
            roof["frac"] = round(roof["achieved"] / HBM_PEAK_GBS, 4)
            roof["timing"] = "HIP events around 260 back-to-back launches on the model's 13 weight pairs (cold)"
    return roof, table


def replay_ffn_in(model, rows, device, rounds=20):
    """The dominant kernel alone: the SwiGLU up-projection on the model's own weights, layer after layer (0.6 GB of
    weights rotate through the caches, so every launch streams from HBM as in the loop), launched back to back
    between ONE pair of HIP events on the current stream -- kernel time plus the dependent-launch boundary, without
    the per-launch event pair of the table above."""
    from lcm import _hip

    sd, layers, i = model.state_dict(), [], 0
    while f"denoiser.layers.{i}.ffn.gate_proj.weight" in sd:
        p = f"denoiser.layers.{i}.ffn."
        layers.append(tuple(sd[p + n] for n in ("gate_proj.weight", "inner_proj.weight", "gate_proj.bias", "inner_proj.bias")))
        i += 1
    if not layers:
        return {}
    F, K = layers[0][0].shape
    A = torch.randn(rows, K, device=device).to(model.dtype)
    out = torch.empty(rows, F, device=device, dtype=model.dtype)
    st, code = _hip.stream_ptr(device), _hip.dtype_code(model.dtype)

    def sweep():
        for g, u, bg, bu in layers:
            _hip.check(_hip.lib.lcmhip_gemm(_hip.ptr(A), K, _hip.ptr(g), _hip.ptr(u), _hip.ptr(bg), _hip.ptr(bu), None,
                                            _hip.ptr(out), F, rows, F, K, _hip.EPI_SWIGLU, code, None, 0, st))
    sweep()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        sweep()
    e1.record()
    e1.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / (rounds * len(layers))
    return {"back_to_back_us": round(us, 3)}


def whole_step_roofline(model, batch, prompt_len, gen_len, seconds_per_job):
    """The whole job against the HBM roof: algorithmic bytes of one executed diffusion step -- every denoiser
    weight once (shared by all rows) plus the cross-attention keys and values of the conditional rows at the mean
    context length -- over the job's wall time per executed diffusion step (which also carries the contextualizer
    and the host loop)."""
    c = model.config
    d, D, F, L = c.model_dim, c.sonar_embed_dim, c.denoiser.allocated_ffn_dim(), c.denoiser.num_layers
    es = 2 if model.dtype == torch.bfloat16 else 4
    n_ffn = 3 if c.denoiser.use_swiglu else 2
    # every GEMM weight a denoiser row passes through: self-attention q, k, v, o + cross-attention q, o + FFN ...
    weights = (2 * D * d + L * (4 * d * d + 2 * d * d + n_ffn * d * F)) * es
    # ... of which this build streams less: the one-position self-attention is folded into a single d x d matrix
    streamed = (2 * D * d + L * (d * d + 2 * d * d + n_ffn * d * F)) * es
    mean_keys = prompt_len + (gen_len - 1) / 2.0 + 1.0                              # zero slot + context
    kv = batch * L * 2 * mean_keys * d * es
    steps = len(model.noise_scheduler.timesteps)
    ms = 1e3 * seconds_per_job / (gen_len * steps)
    achieved = (weights + kv) / ms / 1e6
    return dict(bytes_per_diffusion_step=round(weights + kv), weight_bytes=round(weights), kv_bytes=round(kv),
                streamed_weight_bytes=round(streamed), ms_per_diffusion_step=round(ms, 4), achieved=round(achieved, 1),
                peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 4),
                streamed_frac=round((streamed + kv) / ms / 1e6 / HBM_PEAK_GBS, 4))


def check_timed_output(model, gen, seqs, noise, prompt_len, gen_len, device, doc=0):
    """Cross-check of the timed run (side stream, graph replay, batch-wide tile plans) against an independent run
    on the default stream: document `doc` alone, its first `t` timed output sentences as the prompt, must reproduce
    timed sentence t from the same noise -- once early and once at the longest context."""
    from lcm.datasets.batch import EmbeddingsBatch

    out = {"doc": doc, "positions": [], "cosine": [], "bar": 0.999}
    for t in (prompt_len + 1, prompt_len + gen_len - 1):
        prompt = seqs[doc:doc + 1, :t].to(model.dtype)
        nz = noise[t - prompt_len:t - prompt_len + 1, doc:doc + 1]
        one = gen(EmbeddingsBatch(prompt, None), max_gen_len=1, noise=nz).hypotheses[0][0].seq
        c, s = model.sonar_normalizer.center.float(), model.sonar_normalizer.scale.float()
        a, b = (one[t].float() - c) / s, (seqs[doc, t].float() - c) / s
        out["positions"].append(t)
        out["cosine"].append(round(float(torch.nn.functional.cosine_similarity(a, b, dim=0)), 6))
    torch.cuda.synchronize(device)
    out["pass"] = min(out["cosine"]) >= out["bar"]
    return out


# kernel family -> the epilogue id in the HIP symbol rocprofv3 reports for it
_PMC_SYMBOL = {"den.ffn_in": "k_gemm_dma<3, 4,", "den.norm_mod": "k_norm_modulate", "den.cross_attn": "k_attention"}


def pmc_traffic(family):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (FETCH_SIZE doubled for gfx950 + WRITE_SIZE; counters cannot be read from inside this process)."""
    import glob
    prof = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    found = sorted(glob.glob(os.path.join(prof, "r*_pmc_hbm_traffic.json")))      # the latest round's passes
    path = found[-1] if found else os.path.join(prof, "r1_pmc_hbm_traffic.json")
    rnd = os.path.basename(path).split("_")[0]
    key = _PMC_SYMBOL.get(family)
    if key is None or not os.path.exists(path):
        return {"traffic": None}
    out = {"traffic": None}
    for row in json.load(open(path))["kernels"]:
        if key in row["kernel"]:
            out = {"traffic": row["hbm_bytes_per_launch_corrected"],
                   "traffic_source": f"profiles/{rnd}_pmc_hbm_traffic.json (separate rocprofv3 --pmc passes)"}
            break
    # cross-reference: the same kernel's average duration in the committed rocprofv3 --kernel-trace --stats summary
    # (pure kernel time; the event pair used above also spans the launch boundary, about 2 us)
    stats = os.path.join(os.path.dirname(path), f"{rnd}_bench_kernel_stats.csv")
    if os.path.exists(stats):
        import csv
        for r in csv.DictReader(open(stats, newline="")):
            if key in r["Name"]:
                out["rocprof_avg_us"] = round(float(r["AverageNs"]) / 1e3, 3)
                break
    return out


def cpu_baseline(batch, prompt_len, sentence_steps):
    """The oracle (reference algorithm, PyTorch-CPU fp32) on the same workload shape: the prompt prefill, then
    `sentence_steps` generated sentences per document, each timed on its own; the rate is extrapolated linearly
    from the mean per-sentence time.  Threads = this job's CPU share (OMP_NUM_THREADS on the GPU box, else every
    core the process may run on)."""
    import oracle
    from oracle.model import KVCache
    from oracle.weights import synthetic_noise, synthetic_prompts

    host_cpus = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = host_cpus
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or usable
    torch.set_num_threads(cores)
    cfg = oracle.get_arch(ARCH)
    sd = oracle.init_state_dict(cfg, seed=0, mod_gain=10.0)
    model = oracle.OracleTwoTower(cfg, sd)
    model.prep_for_denoising(oracle.GenOptions(**OPTS))
    x = synthetic_prompts(oracle.load_dummy_normalizer(), batch, prompt_len)
    nz = synthetic_noise(sentence_steps, batch)
    with torch.inference_mode():
        cache = KVCache(prompt_len + sentence_steps)
        t0 = time.perf_counter()
        context = model.encode(model.normalize(x[:, :prompt_len - 1]), cache)      # generator.py:222-259
        cache.increment_step_nr(prompt_len - 1)
        t_prefill = time.perf_counter() - t0
        last, per_step = x[:, prompt_len - 1:prompt_len], []
        for g in range(sentence_steps):
            t0 = time.perf_counter()
            last, context = model.predict_next_sentence(last, context, cache, nz[g].unsqueeze(1))
            per_step.append(time.perf_counter() - t0)
    mean_step = sum(per_step) / len(per_step)
    return dict(value=round(batch / mean_step, 3), unit="sentence-embeddings/s", cores=cores, kind="port",
                host_logical_cpus=host_cpus, prefill_s=round(t_prefill, 2), per_sentence_step_s=[round(t, 2) for t in per_step],
                sample=f"oracle (PyTorch-CPU fp32 restatement of the reference), {batch} docs x {prompt_len} prompt sentences: "
                       f"prefill {t_prefill:.1f} s, then {sentence_steps} generated sentences per doc (context {prompt_len}.."
                       f"{prompt_len + sentence_steps - 1}) at 50 executed DDIM steps + CFG, {mean_step:.1f} s each on {cores} threads "
                       f"(the host shows {host_cpus} logical CPUs); value = docs / mean sentence-step time, i.e. the 128-sentence job "
                       f"extrapolated linearly without its prefill -- the reference re-projects the cross-attention keys/values of "
                       f"the whole context at every diffusion step, so its later sentences are slower than this sample")


def main():
    args = parse()
    launch_ranks(args)          # returns only in a rank whose launcher's world size equals --gpus (or for --gpus 1)
    if args.launcher_selftest:
        return launcher_selftest(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # rehearsal on a one-GPU box: LCM_BENCH_BACKEND=gloo LCM_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # gathers through the host; the real multi-GPU run uses RCCL ("nccl") with one rank per GPU
    backend = os.environ.get("LCM_BENCH_BACKEND", "nccl")
    if os.environ.get("LCM_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()
        assert world == args.gpus, f"--gpus {args.gpus} but the process group has {world} ranks"

    from lcm.datasets.batch import EmbeddingsBatch
    from lcm.inference.two_tower_diffusion_lcm import DiffusionLCMGeneratorOptions, TwoTowerDiffusionLCMGenerator
    from lcm.models.two_tower_diffusion_lcm import create_two_tower_diffusion_lcm_model, lcm_archs
    from lcm.utils.card_utils import load_model_from_card
    from lcm.utils.distributed import all_gather_generations

    B, P, G = args.batch, args.prompt_len, args.gen_len
    opts = dict(OPTS, max_seq_len=max(G, OPTS["max_seq_len"]))
    model = create_two_tower_diffusion_lcm_model(lcm_archs.get_config(args.arch), device=device, dtype=torch.bfloat16, seed=0)
    model.init_parameters(seed=0, modulator_std=0.2)       # non-zero adaLN modulators: every layer does real work
    normalizer = load_model_from_card("dummy_sonar_normalizer", device=device, dtype=torch.bfloat16)
    model.sonar_normalizer.load_state_dict(normalizer.state_dict())
    if args.no_graph:
        from lcm import _hip
        model.session_flags = _hip.SESSION_NO_GRAPH
    gen = TwoTowerDiffusionLCMGenerator(model, DiffusionLCMGeneratorOptions(**opts))

    first_doc = rank * B
    prompts = make_prompts(model, B, P, first_doc, device).to(model.dtype)
    noise = make_noise(G, B, first_doc, device)
    stream = torch.cuda.Stream(device)
    # weights, normalizer buffers, prompts and noise were produced on the default stream
    stream.wait_stream(torch.cuda.current_stream(device))
    torch.cuda.synchronize(device)

    def one_step():
        with torch.cuda.stream(stream):
            out = gen(EmbeddingsBatch(prompts, None), max_gen_len=G, noise=noise)
            seqs = torch.stack([h[0].seq for h in out.hypotheses])           # (B, P+G, D)
            lens = torch.tensor([h[0].seq.shape[0] for h in out.hypotheses], device=device, dtype=torch.int32)
            if distributed:
                if backend == "nccl":
                    seqs, lens = all_gather_generations(seqs, lens, width=B)
                else:
                    seqs, lens = all_gather_generations(seqs.cpu(), lens.cpu(), width=B)
                    seqs, lens = seqs.to(device), lens.to(device)
        return seqs, lens

    def fence():
        torch.cuda.synchronize(device)      # this rank's side-stream work is done before it enters the barrier
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        one_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        seqs, lens = one_step()
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    generated = int((lens.to(torch.int64) - P).sum().item())                 # all ranks' documents after the gather
    assert seqs.shape[0] == world * B and lens.shape[0] == world * B, (tuple(seqs.shape), world, B)   # every rank's documents arrived
    assert generated == world * B * G, (generated, world, B, G)
    assert torch.isfinite(seqs.float()).all()

    result = None
    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        result = {
            "metric": "sentence-embeddings generated/s, 1.6B two-tower LCM, 40 diffusion steps",
            "value": round(generated / (elapsed / args.steps), 2), "unit": "sentence-embeddings/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"config 3: {args.arch}, random-init, {B} docs/GPU x {P} prompt + {G} generated "
                                   f"sentences, 40 requested = 50 executed DDIM steps, CFG w=3 rescale=0.7",
                       "global_batch": world * B, "prompt_len": P, "gen_len": G, "executed_ddim_steps": 50,
                       "parallelism": f"dp{world} by document, one all-gather of outputs"},
        }
    if rank == 0 and not args.no_check:
        result["check"] = check_timed_output(model, gen, seqs[:B], noise, P, G, device)
    if rank == 0 and not args.no_roofline:      # the other ranks wait at the final barrier meanwhile
        with torch.cuda.stream(stream):
            roof, table = roofline_pass(model, TwoTowerDiffusionLCMGenerator, DiffusionLCMGeneratorOptions, B, P, G, device)
        roof["whole_step"] = whole_step_roofline(model, B, P, G, elapsed / args.steps)
        result["roofline"] = roof
        result["kernels"] = table
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(B, P, args.cpu_sentence_steps)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and result.get("check") and not result["check"]["pass"]:
        print("bench.py: the timed output failed its cross-check", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
