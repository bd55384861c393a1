#!/bin/bash
# Runs the end-to-end parity tests once per non-default setting of the measurement-only switches.
# Usage (on the GPU box): bash scripts/switch_matrix.sh > gpurun_out/switch_matrix.log
set -o pipefail
T="tests/test_gpu_generation.py tests/test_gpu_forward.py tests/test_gpu_base_lcm.py"
for sw in LCMHIP_GRAPH=0 LCMHIP_GEMM_DMA=0 LCMHIP_GEMM_DMA=1 LCMHIP_GEMM_NT=0 LCMHIP_TALL=0 LCMHIP_BIG256=0 LCMHIP_NORM_FAST=0; do
    echo "== $sw"
    env "$sw" timeout -k 10 600 python -u -m pytest $T -m gpu -x -q --timeout 120 --timeout-method thread 2>&1 | grep -v amdgpu.ids | tail -3 || exit 1
done
echo "switch matrix: all green"
