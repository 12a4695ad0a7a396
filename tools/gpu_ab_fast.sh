#!/bin/bash
# same-box A/B of the short-time spectral models and the mel front-end: previous build vs current, two repetitions
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-abf}
timeout 600 python -m pytest tests/test_parity_fast.py tests/test_mel.py tests/test_backward_fast.py -m gpu -x -q 2>&1 | tail -3 | tee "$O/${V}_pytest_subset.log"
for rep in 1 2; do
  for m in ${MODELS:-combsubfast combsubsuperfast mel}; do
    env timeout 300 python bench.py --model $m --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_${m}_cur_$rep.json"
    env DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_prev.so timeout 300 python bench.py --model $m --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_${m}_prev_$rep.json"
  done
done
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "abf")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), "%.3e" % d["value"], "kernel_ms", round(d["roofline"]["avg_ms"], 4))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
PY
