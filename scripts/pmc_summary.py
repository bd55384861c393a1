"""Summarises the two rocprofv3 --pmc passes written by scripts/profile_bench.sh <tag> (FETCH_SIZE, WRITE_SIZE; one pass each)
into profiles/r<N>_pmc_hbm_traffic.json (usage: pmc_summary.py r2), and copies the kernel-trace statistics and the bench line measured under the
profiler next to it.  gfx950 correction: FETCH_SIZE counts a 128-byte read request as 64 bytes, so it is doubled;
WRITE_SIZE is exact; both counters are in KB."""
import collections, csv, json, os, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = sys.argv[1] if len(sys.argv) > 1 else "r2"
src = os.path.join(root, "gpurun_out", "prof_" + RND)
dst = os.path.join(root, "profiles")


def per_kernel(path, counter):
    tot, n = collections.Counter(), collections.Counter()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter and r["Kernel_Name"].startswith(("void lcmhip::", "lcmhip::")):
                tot[r["Kernel_Name"]] += float(r["Counter_Value"])
                n[r["Kernel_Name"]] += 1
    return tot, n


ft, fn = per_kernel(os.path.join(src, "pmc_fetch", "bench_counter_collection.csv"), "FETCH_SIZE")
wt, wn = per_kernel(os.path.join(src, "pmc_write", "bench_counter_collection.csv"), "WRITE_SIZE")
rows = []
for k in sorted(ft, key=lambda k: -ft[k]):
    f_avg, w_avg = ft[k] / fn[k], (wt[k] / wn[k] if wn[k] else 0.0)
    rows.append({"kernel": k, "FETCH_SIZE_KB_avg": round(f_avg, 1), "dispatches_fetch": fn[k],
                 "WRITE_SIZE_KB_avg": round(w_avg, 1), "dispatches_write": wn[k],
                 "hbm_bytes_per_launch_corrected": int((2 * f_avg + w_avg) * 1024)})
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (one pass each) -- python3 bench.py --steps 1 --warmup 0 --gen-len 2 "
                  "--no-cpu-baseline --no-roofline   (--no-graph: kernels are launched one by one, the same launches a replayed graph holds)",
       "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B -> doubled; WRITE_SIZE exact; both in KB",
       "kernels": rows}
json.dump(out, open(os.path.join(dst, RND + "_pmc_hbm_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "stats", "bench_kernel_stats.csv"), os.path.join(dst, RND + "_bench_kernel_stats.csv"))
line = [l for l in open(os.path.join(src, "bench_stats.log")) if l.startswith("{")]
if line:
    open(os.path.join(dst, RND + "_bench_under_rocprof.json"), "w").write(line[-1])
print("kernels:", len(rows), "; first:", rows[0]["kernel"][:60], rows[0]["hbm_bytes_per_launch_corrected"])
