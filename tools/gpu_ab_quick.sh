#!/bin/bash
# quick same-box A/B: CombSub step (two streams / one stream) and Sins, previous build vs current; FIR timeline probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-abq}
timeout 600 python -m pytest tests/test_parity.py tests/test_fuzz.py tests/test_backward_fir.py tests/test_noise_rng.py tests/test_fullsize_gpu.py tests/test_core_api.py tests/test_baseline_shapes.py tests/test_modules.py tests/test_sharding.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$O/${V}_pytest_subset.log"
B="python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also"
run() { tag=$1; shift; env "$@" timeout 300 $B 2>&1 | tail -1 > "$O/${V}_bench_$tag.json"; }
for rep in 1 2; do
  run prev_$rep DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_prev.so
  run cur_$rep X=1
  [ -n "${KNOB:-}" ] && run cur_knob_$rep $KNOB
  run prev_one_$rep DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_prev.so DDSP_HIP_ONE_STREAM=1
  run cur_one_$rep DDSP_HIP_ONE_STREAM=1
done
env timeout 300 python bench.py --model sins --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_sins_cur.json"
env DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_prev.so timeout 300 python bench.py --model sins --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_sins_prev.json"
[ -f tools/ab/libddsp_hip_tl.so ] && timeout 120 python tools/${PROBE:-fir_blk_timeline.py} > "$O/${V}_timeline.txt" 2>&1
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "abq")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), round(d.get("ms_per_step_events") or 0, 4), "%.3e" % d["value"], "kernel_ms", round(d["roofline"]["avg_ms"], 4))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
PY
cat "$O/${V}_timeline.txt"
