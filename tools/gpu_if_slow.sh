#!/bin/bash
# appended to other gpurun calls: if this box is one of the slow group (round-2 library's CombSub step above 0.40 ms), collect the
# instruction-fetch counters of tools/gpu_r03_icache.sh here; otherwise do nothing (costs ~8 s)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"
ms=$(DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_prev.so timeout 300 python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --steps 60 2>&1 | tail -1 | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
echo "box check: round-2 library step $ms ms"
if python -c "import sys; sys.exit(0 if float('$ms') > 0.40 else 1)"; then
  V=r03_icache_slow bash tools/gpu_r03_icache.sh > /dev/null 2>&1
  echo "SLOW BOX: counters in gpurun_out/r03_icache_slow_box_compare.txt"
fi
