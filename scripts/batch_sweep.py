"""Runs bench.py at several documents-per-GPU settings (short generation) and prints throughput plus the average
per-launch time of the denoiser's GEMM families.  Usage: python scripts/batch_sweep.py 64 128 256 512"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for b in (sys.argv[1:] or ["64", "128", "256", "512"]):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", b, "--gen-len", "16", "--steps", "1",
                          "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    if out.returncode != 0:
        print(out.stderr[-2000:]); sys.exit(1)
    d = json.loads(out.stdout.strip().splitlines()[-1]); k = d["kernels"]
    fam = ("den.norm_mod", "den.self_ov", "den.cross_q", "den.cross_attn", "den.ffn_in", "den.ffn_out")
    print(f"docs {int(b):4d}  {d['value']:8.1f} sentences/s  {d['ms_per_step']:8.1f} ms  " +
          "  ".join(f"{n[4:]} {k[n]['avg_us']:.1f}" for n in fam), flush=True)
