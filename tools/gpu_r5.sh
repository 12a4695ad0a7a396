#!/bin/bash
# sinusoid bank v3 (amplitude interpolation first), half width folded into the GEMM epilogue, phase sums 4 frames/wave
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_parity.py tests/test_fullsize_gpu.py -m gpu -x -q -k "sins or bank or phase or tail or second_stream or impulse" 2>&1 | tail -4 | tee "$O/pytest_r5.log"
for m in combsub sins; do
  timeout 200 python bench.py --model $m --no-cpu-baseline 2>&1 | tail -1 > "$O/bench_${m}_r5.json"
done
cd /tmp
for m in combsub sins; do
  timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$m" -o $m -- python "$R/bench.py" --model $m --steps 20 --warmup 3 --prewarm-seconds 0.2 --no-cpu-baseline > "$O/prof_$m.log" 2>&1
  python "$R/tools/rocpd_stats.py" $(find "$O/prof_$m" -name "*.db" | head -1) 2>&1 | head -16 | tee "$O/${m}_kernel_stats.csv"
  rm -rf "$O/prof_$m"
done
cd "$R"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/bench_*_r5.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"], 4), "%.3e" % d["value"])
PY
