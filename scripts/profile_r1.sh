#!/bin/bash
# Profile of the default bench command: kernel-trace stats, then two PMC passes on a short run.
# The counter passes replay nothing from a graph (LCMHIP_GRAPH=0): one pass with graph replay stopped
# responding right after start-up.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r1
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_stats.log 2>&1
find $OUT/stats -name '*kernel_trace.csv' -size +20M -delete
echo stats done
LCMHIP_GRAPH=0 timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --gen-len 2 --no-cpu-baseline --no-roofline > $OUT/bench_fetch.log 2>&1
echo fetch done
LCMHIP_GRAPH=0 timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --gen-len 2 --no-cpu-baseline --no-roofline > $OUT/bench_write.log 2>&1
echo write done
ls -laR $OUT | head -50
