#!/bin/bash
# Matrix-core counters of the many-row GEMM kernels at 4096 rows (scripts/gemm_big_bench.py): one kernel-trace pass for
# durations, one pass with SQ counters, one with GRBM_GUI_ACTIVE (the clock the chip held); summarised per kernel.
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_mfma
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GB_NO_TORCH=1 GB_ROWS=4096 GB_ITERS=10
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o gb -- python3 $R/scripts/gemm_big_bench.py > $OUT/stats.log 2>&1 || { echo "stats run failed"; tail -5 $OUT/stats.log; exit 1; }
echo stats done
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT/pmc_sq -o gb -- python3 $R/scripts/gemm_big_bench.py > $OUT/sq.log 2>&1 || { echo "sq pass failed"; tail -5 $OUT/sq.log; exit 1; }
echo sq done
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_grbm -o gb -- python3 $R/scripts/gemm_big_bench.py > $OUT/grbm.log 2>&1 || { echo "grbm pass failed"; tail -5 $OUT/grbm.log; exit 1; }
echo grbm done
ls -R $OUT | head -40
