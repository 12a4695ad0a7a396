#!/bin/bash
# second-stream noise branch: parity at test and full size, step time with / without it (clock ramp-up included in bench.py)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_parity.py tests/test_fullsize_gpu.py -m gpu -x -q -k "second_stream" 2>&1 | tail -4 | tee "$O/pytest_r3.log"
for m in combsub sins; do
  timeout 200 python bench.py --model $m --no-cpu-baseline 2>&1 | tail -1 > "$O/bench_${m}_2s.json"
  DDSP_HIP_ONE_STREAM=1 timeout 200 python bench.py --model $m --no-cpu-baseline 2>&1 | tail -1 > "$O/bench_${m}_1s.json"
  timeout 200 python bench.py --model $m --no-cpu-baseline --prewarm-seconds 0 --steps 20 --warmup 3 2>&1 | tail -1 > "$O/bench_${m}_cold.json"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_*_2s.json") + glob.glob("gpurun_out/bench_*_1s.json") + glob.glob("gpurun_out/bench_*_cold.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"], 4), "%.3e" % d["value"])
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
PY
