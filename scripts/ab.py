"""A/B helper: runs bench.py under several environment variants and prints value + per-kernel averages.
Usage: python scripts/ab.py "NAME:VAR=1,VAR2=3" ...   (first variant 'base:' = no overrides)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gen = os.environ.get("AB_GEN_LEN", "24")
for spec in sys.argv[1:]:
    name, _, kv = spec.partition(":")
    env = dict(os.environ)
    for item in filter(None, kv.split(",")):
        k, v = item.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gen-len", gen, "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, "FAILED", r.stderr[-2000:]); sys.exit(1)
    d = json.loads(line[-1])
    ks = d.get("kernels", {})
    cols = ["den.norm_mod", "den.self_ov", "den.cross_q", "den.cross_attn", "den.cross_o", "den.ffn_in", "den.ffn_out"]
    print(f"{name:28s} value {d['value']:8.1f}  " + "  ".join(f"{c[4:]} {ks[c]['avg_us']:.1f}" for c in cols if c in ks), flush=True)
