#!/bin/bash
# round 3, call N: GPU suite, the default bench (with also.rssloss), training steps with and without the real loss, race probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03n}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$O/${V}_pytest_gpu.log"
( time timeout 400 python bench.py ) 2>&1 | tail -5 > "$O/${V}_bench_combsub.log"; grep '^{' "$O/${V}_bench_combsub.log" | tail -1 > "$O/${V}_bench_combsub.json"
for k in combsub sins combsubsuperfast; do
  timeout 120 python tools/train_step_probe.py $k 2>&1 | tail -1
  timeout 120 python tools/train_step_probe.py $k loss 2>&1 | tail -1
done | tee "$O/${V}_train_ms.txt"
timeout 400 python tools/race_probe.py 10 2>&1 | tail -40 > "$O/${V}_race_probe.txt"; grep -c "0 mismatches" "$O/${V}_race_probe.txt"; grep -v "0 mismatches" "$O/${V}_race_probe.txt" | tail -5
python - <<'PY'
import json, os
V = os.environ.get("V", "r03n")
d = json.loads(open("gpurun_out/%s_bench_combsub.json" % V).read())
print("headline ms", round(d["ms_per_step"], 4), "frac", round(d["roofline"]["frac"], 4), "kernel ms", round(d["roofline"]["avg_ms"], 4))
for k, v in d.get("also", {}).items():
    print("also", k, round(v["ms_per_step"], 4))
PY
tail -3 "$O/${V}_bench_combsub.log"
