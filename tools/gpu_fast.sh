#!/bin/bash
# One gpurun call for the CombSubFast / CombSubSuperFast path: GPU parity tests, bench lines, kernel stats.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O="$R/gpurun_out"
mkdir -p "$O"
export TMPDIR=/tmp
echo "== pytest -m gpu =="
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$O/pytest_gpu.log"
for m in combsubsuperfast combsubfast; do
  echo "== bench $m =="
  timeout 600 python bench.py --model $m --steps 10 --warmup 3 ${BENCH_EXTRA:-} 2>&1 | tail -2 | tee "$O/bench_$m.json"
done
echo "== rocprof stats (combsubsuperfast) =="
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_superfast" -o superfast -- python "$R/bench.py" --model combsubsuperfast --steps 5 --warmup 2 --no-cpu-baseline > "$O/prof_superfast.log" 2>&1
python "$R/tools/rocpd_stats.py" $(find "$O/prof_superfast" -name "*.db" | head -1) 2>&1 | head -20 | tee "$O/superfast_kernel_stats.txt"
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_fast" -o fast -- python "$R/bench.py" --model combsubfast --steps 5 --warmup 2 --no-cpu-baseline > "$O/prof_fast.log" 2>&1
python "$R/tools/rocpd_stats.py" $(find "$O/prof_fast" -name "*.db" | head -1) 2>&1 | head -20 | tee "$O/fast_kernel_stats.txt"
