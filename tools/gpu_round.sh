#!/bin/bash
# One gpurun call: GPU parity tests, smoke, headline bench, rocprofv3 kernel stats.  Outputs under gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O="$R/gpurun_out"
mkdir -p "$O"
export TMPDIR=/tmp
echo "== host =="; nproc; lscpu | grep -m1 "Model name"; rocm-smi --showproductname 2>/dev/null | head -5
echo "== pytest -m gpu =="
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$O/pytest_gpu.log"
echo "== smoke =="
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -5 | tee "$O/smoke.log"
echo "== bench combsub =="
timeout 600 python bench.py --steps 10 --warmup 3 2>&1 | tail -3 | tee "$O/bench_combsub.json"
echo "== bench sins =="
timeout 600 python bench.py --model sins --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -3 | tee "$O/bench_sins.json"
echo "== rocprof stats (combsub) =="
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_combsub" -o combsub -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$O/prof_combsub.log" 2>&1
find "$O/prof_combsub" -name "*kernel_stats*" | head -3
for f in $(find "$O/prof_combsub" -name "*kernel_stats.csv" | head -1); do head -25 "$f"; done
echo "== rocprof stats (sins) =="
timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_sins" -o sins -- python "$R/bench.py" --model sins --steps 3 --warmup 1 --no-cpu-baseline > "$O/prof_sins.log" 2>&1
for f in $(find "$O/prof_sins" -name "*kernel_stats.csv" | head -1); do head -25 "$f"; done
