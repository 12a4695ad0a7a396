#!/bin/bash
# same-box A/B of the CombSub step and the FIR kernel alone: previous build (tools/ab/libddsp_hip_prev.so) vs current
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-ab3}
timeout 900 python -m pytest tests/test_parity.py tests/test_parity_fast.py tests/test_mel.py tests/test_fullsize_gpu.py tests/test_fuzz.py tests/test_backward_fir.py tests/test_backward_fast.py -m gpu -x -q 2>&1 | tail -4 | tee "$O/${V}_pytest_subset.log"
B="python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also"
run() { tag=$1; shift; env "$@" timeout 300 $B 2>&1 | tail -1 > "$O/${V}_bench_$tag.json"; }
for rep in 1 2; do
  run prev_$rep DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_prev.so
  run cur_$rep X=1
  run prev_one_$rep DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_prev.so DDSP_HIP_ONE_STREAM=1
  run cur_one_$rep DDSP_HIP_ONE_STREAM=1
done
for m in sins combsubsuperfast combsubfast mel; do
  env timeout 300 python bench.py --model $m --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_${m}_cur.json"
  env DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_prev.so timeout 300 python bench.py --model $m --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_${m}_prev.json"
done
cd /tmp
DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_ab3" -o ab -- python "$R/bench.py" --only-steps --steps 20 --warmup 3 > "$O/prof_ab3.log" 2>&1
python "$R/tools/rocpd_stats.py" $(find "$O/prof_ab3" -name "*.db" | head -1) 2>&1 | head -9 > "$O/${V}_cur_kernel_stats.csv"
rm -rf "$O/prof_ab3"
cd "$R"
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "ab3")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), round(d.get("ms_per_step_events") or 0, 4), "%.3e" % d["value"], "kernel_ms", round(d["roofline"]["avg_ms"], 4))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
PY
cat "$O/${V}_cur_kernel_stats.csv"
