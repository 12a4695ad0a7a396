#!/bin/bash
# round 3, call H: backward-FIR GPU tests, adjoint kernel timing prev vs cur, race probe of the adjoint, training-step traces
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03h}
timeout 300 python -m pytest tests/test_backward_fir.py tests/test_fullsize_gpu.py tests/test_core_api.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee "$O/${V}_pytest_subset.log"
echo "== adjoint kernel alone, prev:"; DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_prev.so timeout 120 python tools/fir_bwd_bench.py 2>&1 | tail -2
echo "== adjoint kernel alone, cur:"; timeout 120 python tools/fir_bwd_bench.py 2>&1 | tail -2
MODELS="combsub sins" PROF_TIMEOUT=60 V=${V} bash tools/gpu_train_prof.sh 2>&1 | cut -c1-150
