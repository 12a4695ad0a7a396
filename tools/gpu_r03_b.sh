#!/bin/bash
# round 3, call B: FIR-related GPU tests, same-box A/B (prev = round-2 kernel), per-workgroup timeline, SQ counters of the CombSub step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03b} bash tools/gpu_ab_quick.sh
MODELS=combsub bash tools/gpu_step_pmc.sh > /dev/null 2>&1
cp "$O/step_pmc_combsub.txt" "$O/${V:-r03b}_step_pmc_combsub.txt"
grep "k_fir_blk" "$O/step_pmc_combsub.txt"
