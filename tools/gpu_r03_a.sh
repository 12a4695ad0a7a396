#!/bin/bash
# round 3, call A: issue-rate probe 2, FIR-related GPU tests, same-box A/B of the filter rewrite (prev = round-2 kernel)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 120 tools/ab/valu_probe2.bin > "$O/r03_valu_probe2.txt" 2>&1
V=${V:-r03a} bash tools/gpu_ab_quick.sh
