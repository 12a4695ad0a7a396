#!/bin/bash
# round 3, call G: SQ counters of the Sins step's kernels, round-2 library against the current one
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03g}
for tag in prev cur; do
  if [ $tag = cur ]; then unset DDSP_HIP_LIB; else export DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so; fi
  MODELS=sins bash tools/gpu_step_pmc.sh > /dev/null 2>&1
  cp "$O/step_pmc_sins.txt" "$O/${V}_step_pmc_sins_$tag.txt"
  echo "== $tag"; grep "k_sins_bank2" "$O/step_pmc_sins.txt"
done
