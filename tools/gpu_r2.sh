#!/bin/bash
# round-1 late batch: spectral loss parity + bench, pipelined tap-synthesis GEMM (k_ir_gemm_p) vs the one-tile form, two-stream probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_loss.py tests/test_parity.py -m gpu -x -q -k "loss or impulse or tile_runs or tail or full_size" 2>&1 | tail -5 | tee "$O/pytest_r2.log"
timeout 120 python tools/gemm_probe.py 2>&1 | tail -40 > "$O/gemm_new.json"
DDSP_HIP_GEMM_V1=1 timeout 120 python tools/gemm_probe.py 2>&1 | tail -40 > "$O/gemm_v1.json"
for per in 1 4; do DDSP_HIP_GEMM_PER=$per timeout 120 python tools/gemm_probe.py 2>&1 | tail -40 > "$O/gemm_per$per.json"; done
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > "$O/bench_combsub_new.json"
DDSP_HIP_GEMM_V1=1 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > "$O/bench_combsub_v1.json"
timeout 200 python bench.py --model sins --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 > "$O/bench_sins_new.json"
timeout 120 python tools/stream_probe.py 2>&1 | tail -20 > "$O/stream_probe.json"
timeout 200 python bench.py --model rssloss --steps 10 --warmup 2 2>&1 | tail -1 > "$O/bench_rssloss.json"
grep -h "ms_per_step" "$O"/bench_combsub_new.json "$O"/bench_combsub_v1.json "$O"/bench_sins_new.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['metric'][:40], d['ms_per_step'])
"
cat "$O/stream_probe.json"
python - <<'PY'
import json
for f in ("gemm_new", "gemm_v1", "gemm_per1", "gemm_per4"):
    try:
        t = open("gpurun_out/%s.json" % f).read()
        i = t.index("{"); a = json.loads(t[i:t.index("}") + 1])
        print(f, {k: v for k, v in a.items() if "rows27584" in k})
    except Exception as e:
        print(f, "ERR", e)
PY
cat "$O/bench_rssloss.json" | cut -c1-1500
