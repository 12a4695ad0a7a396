#!/bin/bash
# Round refresh in one gpurun call: full GPU suite, smoke, every bench line, per-kernel stats (one-stream order so
# durations are not inflated by overlap), HBM traffic PMC passes.  Outputs under gpurun_out/, copied to profiles/ by hand.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-v10}
echo "== host =="; nproc; lscpu | grep -m1 "Model name"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$O/${V}_pytest_gpu.log"
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -4 | tee "$O/${V}_smoke.log"
timeout 600 python bench.py 2>&1 | tail -1 > "$O/${V}_bench_combsub.json"
for m in sins combsubfast combsubsuperfast mel sinesrc rssloss; do
  timeout 300 python bench.py --model $m --no-cpu-baseline 2>&1 | tail -1 > "$O/${V}_bench_$m.json"
done
DDSP_HIP_ONE_STREAM=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 > "$O/${V}_bench_combsub_one_stream.json"
DDSP_HIP_ONE_STREAM=1 timeout 300 python bench.py --model sins --no-cpu-baseline 2>&1 | tail -1 > "$O/${V}_bench_sins_one_stream.json"
cd /tmp
for m in combsub sins; do
  DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$m" -o $m -- python "$R/bench.py" --model $m --steps 20 --warmup 3 --prewarm-seconds 0.2 --no-cpu-baseline > "$O/prof_$m.log" 2>&1
  python "$R/tools/rocpd_stats.py" $(find "$O/prof_$m" -name "*.db" | head -1) 2>&1 | head -14 > "$O/${V}_${m}_kernel_stats.csv"
  rm -rf "$O/prof_$m"
done
cd "$R"
DDSP_HIP_ONE_STREAM=1 BENCH_ARGS="--prewarm-seconds 0" bash tools/gpu_traffic.sh > "$O/${V}_traffic.log" 2>&1
cp "$O/traffic.json" "$O/${V}_hbm_traffic.json"; cp "$O/traffic_FETCH_SIZE.txt" "$O/${V}_pmc_FETCH_SIZE.txt"; cp "$O/traffic_WRITE_SIZE.txt" "$O/${V}_pmc_WRITE_SIZE.txt"
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "v10")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), "%.3e" % d["value"], d["unit"])
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-400:])
PY
head -12 "$O/${V}_combsub_kernel_stats.csv"; head -10 "$O/${V}_sins_kernel_stats.csv"; cat "$O/${V}_hbm_traffic.json" | head -30
