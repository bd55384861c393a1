#!/bin/bash
# Round-2 profile of the default bench command: rocprofv3 kernel-trace statistics, two PMC passes (HBM traffic) on a
# short run, then a ladder of graph sizes under a PMC pass (how many kernel nodes a replayed graph may have before a
# counter-collecting profiler stalls).  Every GPU step has its own time limit; the script stops at the first failure.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_r2
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_stats.log 2>&1 || { echo "stats run failed"; tail -5 $OUT/bench_stats.log; exit 1; }
find $OUT/stats -name '*kernel_trace.csv' -size +20M -delete
echo stats done
LCMHIP_GRAPH=0 timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --gen-len 2 --no-cpu-baseline --no-roofline --no-check > $OUT/bench_fetch.log 2>&1 || { echo "fetch pass failed"; exit 1; }
echo fetch done
LCMHIP_GRAPH=0 timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --gen-len 2 --no-cpu-baseline --no-roofline --no-check > $OUT/bench_write.log 2>&1 || { echo "write pass failed"; exit 1; }
echo write done
# ---- graph replay under counter collection: growing graphs, stop at the first one that does not finish
export LCMHIP_DEBUG_GRAPH=1
for steps in 5 20 40 100; do
  timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_graph_$steps -o probe -- python3 $R/scripts/pmc_graph_probe.py $steps > $OUT/graph_probe_$steps.log 2>&1
  rc=$?
  echo "graph probe, $steps requested steps: rc=$rc; $(grep -c 'lcmhip graph' $OUT/graph_probe_$steps.log) debug lines; last: $(grep 'lcmhip graph\|probe ok' $OUT/graph_probe_$steps.log | tail -1)"
  rm -rf $OUT/pmc_graph_$steps
  if [ $rc -ne 0 ]; then break; fi
done
ls $OUT
