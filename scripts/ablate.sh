#!/bin/bash
# runs scripts/gemm_bench.py against every ablation build, plain and with the tall tile
for n in "" d1 ab1 ab2 ab3 ab3d2; do
  for tall in 0 1; do
    lib=$PWD/chimera-llm_amd/lcm/liblcm_hip.so; [ -n "$n" ] && lib=$PWD/chimera-llm_amd/lcm/ablate/liblcm_hip_$n.so
    echo "== ${n:-default} tall=$tall"
    LCMHIP_LIB=$lib LCMHIP_TALL=$tall timeout -k 10 120 python scripts/gemm_bench.py 2>/dev/null || exit 1
  done
done
