#!/bin/bash
# round 3, call Q: same-box A/B of the short-time spectral models: tools/ab/libddsp_hip_head.so (the commit before) against the tree
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
for m in combsubsuperfast combsubfast; do
  for tag in head cur head cur; do
    if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
    env $lib timeout 200 python bench.py --model $m --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --steps 100 2>&1 | tail -1 > "$O/q.json"
    python - "$m" "$tag" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/q.json").read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "step ms %.4f" % d["ms_per_step"], "kernel alone %.4f" % d["roofline"]["avg_ms"])
PY
  done
done
for tag in head cur; do
  if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
  env $lib timeout 120 python tools/train_step_probe.py combsubsuperfast 2>&1 | tail -1
done
