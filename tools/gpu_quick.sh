#!/bin/bash
# quick GPU iteration: parity tests + per-kernel timings + headline bench (no profiler)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$O/pytest_gpu.log"
timeout 300 python tools/kernel_bench.py 2>&1 | tail -30 | tee "$O/kernel_bench.json"
timeout 600 python bench.py --steps 10 --warmup 3 ${BENCH_ARGS:-} 2>&1 | tail -2 | tee "$O/bench_combsub.json"
