#!/bin/bash
# round 3, call S: the last pass's LDS reads of the 2048- and 4096-point plans spread over the banks: head against the tree for every
# kernel on those plans (loss, short-time spectral filter, log-mel), then the loss kernels' conflict counters
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
for m in rssloss combsubsuperfast mel; do
  for tag in head cur head cur; do
    if [ $tag = cur ]; then unset DDSP_HIP_LIB; else export DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so; fi
    timeout 200 python bench.py --model $m --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --steps 80 2>&1 | tail -1 > "$O/s.json"
    python - "$m" "$tag" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/s.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "step ms %.4f" % d["ms_per_step"])
PY
  done
done
unset DDSP_HIP_LIB
bash tools/gpu_loss_pmc.sh 2>&1 | grep -E "BANK_CONFLICT|IDX_ACTIVE"
