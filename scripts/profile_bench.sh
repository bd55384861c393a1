#!/bin/bash
# Profile of the default bench command: rocprofv3 kernel-trace statistics, then two PMC passes (HBM traffic) on a
# short run that launches the sampling loop kernel by kernel (the same launches a replayed graph holds).
# Every GPU step has its own time limit; the script stops at the first failure.
# usage (on the GPU box): bash scripts/profile_bench.sh [tag]     -> gpurun_out/prof_<tag>/
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_${1:-r3}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_stats.log 2>&1 || { echo "stats run failed"; tail -5 $OUT/bench_stats.log; exit 1; }
find $OUT/stats -name '*kernel_trace.csv' -size +20M -delete
echo stats done
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --gen-len 2 --no-graph --no-cpu-baseline --no-roofline --no-check > $OUT/bench_fetch.log 2>&1 || { echo "fetch pass failed"; exit 1; }
echo fetch done
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --gen-len 2 --no-graph --no-cpu-baseline --no-roofline --no-check > $OUT/bench_write.log 2>&1 || { echo "write pass failed"; exit 1; }
echo write done
ls $OUT
