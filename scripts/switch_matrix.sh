#!/bin/bash
# Runs the end-to-end parity tests once per non-default setting of the library's environment switches (the kernels
# the default routing does not take must give the same answers).  The session options (LCMHIP_SESSION_* flags) are
# exercised by the tests themselves.
# Usage (on the GPU box): bash scripts/switch_matrix.sh > gpurun_out/switch_matrix.log
set -o pipefail
T="tests/test_gpu_generation.py tests/test_gpu_forward.py tests/test_gpu_base_lcm.py"
n=0
for sw in LCMHIP_BIG256=0 LCMHIP_BIG256_N128=0 "LCMHIP_BIG256=0 LCMHIP_BIG256_N128=0"; do
    echo "== $sw"
    n=$((n + 1))
    # the whole output of a run is kept (a crash is diagnosed from it); only its last lines are echoed
    env $sw timeout -k 10 900 python -u -X faulthandler -m pytest $T -m gpu -x -q --timeout 300 --timeout-method thread \
        > gpurun_out/switch_matrix_run$n.log 2>&1
    rc=$?
    grep -v amdgpu.ids gpurun_out/switch_matrix_run$n.log | tail -3
    [ $rc -eq 0 ] || { echo "run $n failed with status $rc: see gpurun_out/switch_matrix_run$n.log"; exit 1; }
done
echo "switch matrix: all green"
