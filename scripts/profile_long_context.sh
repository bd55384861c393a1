#!/bin/bash
# Long-context configuration (512 documents at context ~500) under rocprofv3: kernel-trace statistics, then HBM read
# traffic (FETCH_SIZE) of the cross-attention kernel, on a bounded sample (2 generated sentences, 10 DDIM steps).
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_long
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export LC_GEN=2 LC_STEPS=10 LC_BATCH=${LC_BATCH:-128}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o lc -- python3 $R/scripts/bench_long_context.py > $OUT/stats.log 2>&1 || { echo "stats run failed"; tail -5 $OUT/stats.log; exit 1; }
find $OUT/stats -name '*kernel_trace.csv' -size +20M -delete
echo stats done
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o lc -- python3 $R/scripts/bench_long_context.py > $OUT/fetch.log 2>&1 || { echo "fetch pass failed"; tail -5 $OUT/fetch.log; exit 1; }
echo fetch done
python3 - <<PY
import csv, collections, glob
tot, n = collections.Counter(), collections.Counter()
for f in glob.glob("$OUT/pmc_fetch/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        if r["Counter_Name"] == "FETCH_SIZE":
            k = r["Kernel_Name"][:70]
            tot[k] += float(r["Counter_Value"]); n[k] += 1
with open("$OUT/fetch_summary.txt", "w") as o:
    for k in sorted(tot, key=lambda k: -tot[k])[:12]:
        o.write(f"{k:70s} launches {n[k]:6d}  FETCH_SIZE avg {tot[k] / n[k]:12.1f} KB  -> x2 (gfx950) = {2 * tot[k] / n[k] / 1e6:8.3f} GB per launch\n")
print(open("$OUT/fetch_summary.txt").read())
PY
find $OUT/pmc_fetch -name '*.csv' -size +5M -delete
ls $OUT $OUT/stats/* | head -30
