"""Summarises scripts/profile_mfma.sh (gpurun_out/prof_mfma) into profiles/<round>_mfma_counters.txt: per kernel and grid
size the matrix-core busy share (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles)), the clock the chip held
(GRBM_GUI_ACTIVE / 8 XCDs / duration) and the rate.  usage: mfma_summary.py r2"""
import collections, csv, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r2"
src = os.path.join(root, "gpurun_out", "prof_mfma")


def key(r):
    return (r["Kernel_Name"].split("(")[0].replace("void lcmhip::", ""), int(r["Grid_Size"]))


sq = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(os.path.join(src, "pmc_sq", "gb_counter_collection.csv"), newline="")):
    if "lcmhip::k_gemm" in r["Kernel_Name"]:
        sq[key(r)][r["Counter_Name"]].append(float(r["Counter_Value"]))
gui, dur = collections.defaultdict(list), collections.defaultdict(list)
for r in csv.DictReader(open(os.path.join(src, "pmc_grbm", "gb_counter_collection.csv"), newline="")):
    if "lcmhip::k_gemm" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        gui[key(r)].append(float(r["Counter_Value"]))
        dur[key(r)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
lines = ["kernel, grid size (threads): MFMA instructions (16 cycles each), kernel cycles = GRBM_GUI_ACTIVE / 8, clock, "
         "matrix-core busy share of 1024 SIMDs, TFLOP/s under the profiler"]
for k in sorted(sq, key=lambda k: (k[0], k[1])):
    avg = lambda v: sum(v) / len(v)          # noqa: E731
    busy = avg(sq[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
    cyc = avg(gui[k]) / 8.0
    ns = avg(dur[k])
    n_mfma = busy / 16.0
    lines.append(f"{k[0]:22s} grid {k[1]:8d}: {n_mfma / 1e6:8.2f} M MFMA, {cyc / 1e3:7.1f} k cycles in {ns / 1e3:6.1f} us = {cyc / ns:4.2f} GHz, "
                 f"busy {busy / (1024.0 * cyc):5.3f}, {n_mfma * 16384 / ns / 1e3:7.1f} TFLOP/s")
open(os.path.join(root, "profiles", rnd + "_mfma_counters.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
