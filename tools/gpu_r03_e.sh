#!/bin/bash
# round 3, call E: the driver's own command (default bench.py: headline + also + live traffic + CPU baselines), the --cfg4 path on
# one GPU with a 1-rank RCCL communicator, and the kernel timeline of one steady-state step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03e}
( time timeout 600 python bench.py ) > "$O/${V}_bench_default.log" 2>&1
tail -5 "$O/${V}_bench_default.log" | cut -c1-3000
timeout 300 python bench.py --cfg4 --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --steps 30 2>&1 | tail -1 > "$O/${V}_bench_cfg4_1rank.json"
python -c "
import json; d=json.loads(open('$O/${V}_bench_cfg4_1rank.json').read().strip().splitlines()[-1]); print('cfg4:', json.dumps(d.get('cfg4'))[:1200])"
V=$V bash tools/gpu_step_gaps.sh 2>&1 | tail -30
