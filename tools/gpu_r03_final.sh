#!/bin/bash
# round 3, last call: the GPU suite and smoke on the final tree, the driver's bench command, the loss and the training steps
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03_v35}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$O/${V}_pytest_gpu.log"
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -4 | tee "$O/${V}_smoke.log"
( time timeout 400 python bench.py ) 2>&1 | tail -5 > "$O/${V}_bench_combsub.log"; grep '^{' "$O/${V}_bench_combsub.log" | tail -1 > "$O/${V}_bench_combsub.json"
for m in sins combsubsuperfast combsubfast rssloss; do
  timeout 300 python bench.py --model $m --no-also --no-cpu-baseline 2>&1 | tail -1 > "$O/${V}_bench_$m.json"
done
for k in combsub sins combsubsuperfast combsubfast; do timeout 120 python tools/train_step_probe.py $k 2>&1 | tail -1; done | tee "$O/${V}_train_ms.txt"
for k in combsub sins combsubsuperfast; do timeout 120 python tools/train_step_probe.py $k loss 2>&1 | tail -1; done | tee -a "$O/${V}_train_ms.txt"
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "r03_v35")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), "%.3e" % d["value"], d["unit"], "frac", round(d.get("roofline", {}).get("frac", 0), 4))
    except Exception as e:
        print(f, "ERR", e)
d = json.loads(open("gpurun_out/%s_bench_combsub.json" % V).read())
print("also", {k: round(v["ms_per_step"], 4) for k, v in d.get("also", {}).items()})
PY
tail -3 "$O/${V}_bench_combsub.log"
