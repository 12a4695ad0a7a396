#!/bin/bash
# fast-path follow-up: GPU parity of the fast tests, variant timings, bench lines, HBM traffic PMC passes (superfast)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_fast.py tests/test_modules.py -m gpu -x -q 2>&1 | tail -5 | tee "$O/pytest_fast_gpu.log"
RUNS="${RUNS:-0 10 14}" python tools/stft_bench.py 2>&1 | tail -30 | tee "$O/stft_variants.json"
for m in combsubsuperfast combsubfast; do
  timeout 600 python bench.py --model $m --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee "$O/bench_$m.json"
done
BENCH_ARGS="--model combsubsuperfast" bash tools/gpu_traffic.sh
cp "$O/traffic.json" "$O/traffic_superfast.json"
