#!/bin/bash
# one call that characterises the box (faster / slower group) and tries what might help the slower ones
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-slow}
{
  echo "== translation probe"; timeout 60 tools/ab/tlb_probe.bin
  echo "== timeline, default map"; timeout 100 python tools/fir_blk_timeline.py 2>&1 | grep -E "launch|prologue|pair  0|pair  1:|end  "
  echo "== first-launch effect"; timeout 200 python tools/fir_cold_start.py
  timeout 200 python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms', round(d['ms_per_step'],4), 'fir', round(d['roofline']['avg_ms'],4))"
} 2>&1 | tee "$O/${V}_slow_box.txt"
