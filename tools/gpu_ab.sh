#!/bin/bash
# quick same-box A/B in one gpurun call: GPU parity subset, then the CombSub step with the tap synthesis in its
# prime-factor form (default) against the dense contraction (DDSP_HIP_TAPS_GEMM=1), each with a one-stream kernel trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-ab}
timeout 900 python -m pytest tests/test_parity.py tests/test_baseline_shapes.py tests/test_core_api.py tests/test_fullsize_gpu.py tests/test_cascade_seam.py tests/test_sharding.py -m gpu -x -q 2>&1 | tail -5 | tee "$O/${V}_pytest_subset.log"
for rep in $(seq 1 ${REPS:-2}); do
  for g in 0 1; do
    DDSP_HIP_TAPS_GEMM=$g timeout 300 python bench.py --no-cpu-baseline --no-module-mode 2>&1 | tail -1 > "$O/${V}_bench_combsub_gemm${g}_$rep.json"
    DDSP_HIP_ONE_STREAM=1 DDSP_HIP_TAPS_GEMM=$g timeout 300 python bench.py --no-cpu-baseline --no-module-mode 2>&1 | tail -1 > "$O/${V}_bench_combsub_gemm${g}_one_stream_$rep.json"
  done
done
DDSP_HIP_TAPS_GEMM=0 timeout 300 python bench.py --model sins --no-cpu-baseline --no-module-mode 2>&1 | tail -1 > "$O/${V}_bench_sins_gemm0.json"
cd /tmp
for g in 0 1; do
  DDSP_HIP_ONE_STREAM=1 DDSP_HIP_TAPS_GEMM=$g timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_ab$g" -o ab -- python "$R/bench.py" --only-steps --steps 20 --warmup 3 > "$O/prof_ab$g.log" 2>&1
  python "$R/tools/rocpd_stats.py" $(find "$O/prof_ab$g" -name "*.db" | head -1) 2>&1 | head -12 > "$O/${V}_gemm${g}_kernel_stats.csv"
  rm -rf "$O/prof_ab$g"
done
cd "$R"
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "ab")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), d.get("ms_per_step_events"), "%.3e" % d["value"])
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
PY
head -12 "$O/${V}_gemm0_kernel_stats.csv"; head -12 "$O/${V}_gemm1_kernel_stats.csv"
