#!/bin/bash
# Round-2 refresh in one gpurun call: GPU suite, smoke, the headline bench line (+ Sins, one-stream order, gather on a
# 1-rank RCCL communicator, cfg-5 seam, --gpus 2 refusal), per-kernel stats (one-stream order so durations are not
# inflated by overlap) and the HBM-traffic PMC passes.  Outputs under gpurun_out/, the judged ones copied to profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r02_v1}
echo "== host =="; nproc; lscpu | grep -m1 "Model name"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$O/${V}_pytest_gpu.log"
  timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -4 | tee "$O/${V}_smoke.log"
fi
timeout 600 python bench.py 2>&1 | tail -1 > "$O/${V}_bench_combsub.json"
timeout 600 python bench.py --model sins 2>&1 | tail -1 > "$O/${V}_bench_sins.json"
timeout 300 python bench.py --gather --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_combsub_gather_1rank.json"
timeout 300 python bench.py --gpus 2 --no-cpu-baseline > "$O/${V}_bench_gpus2_on_1gpu.log" 2>&1; echo "exit $?" >> "$O/${V}_bench_gpus2_on_1gpu.log"
timeout 600 python bench.py --model cascade_seam --batch-per-gpu 64 2>&1 | tail -1 > "$O/${V}_bench_cascade_seam.json"
DDSP_HIP_ONE_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_combsub_one_stream.json"
if [ "${ALL_MODELS:-0}" = "1" ]; then
  for m in combsubfast combsubsuperfast mel sinesrc rssloss; do
    timeout 300 python bench.py --model $m --no-cpu-baseline 2>&1 | tail -1 > "$O/${V}_bench_$m.json"
  done
fi
cd /tmp
for m in combsub sins; do
  DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$m" -o $m -- python "$R/bench.py" --model $m --only-steps --steps 20 --warmup 3 > "$O/prof_$m.log" 2>&1
  python "$R/tools/rocpd_stats.py" $(find "$O/prof_$m" -name "*.db" | head -1) 2>&1 | head -16 > "$O/${V}_${m}_kernel_stats.csv"
  rm -rf "$O/prof_$m"
done
cd "$R"
for m in combsub sins; do
  DDSP_HIP_ONE_STREAM=1 MODEL=$m bash tools/gpu_traffic.sh > "$O/${V}_traffic_$m.log" 2>&1
  cp "$O/traffic.json" "$O/${V}_${m}_hbm_traffic.json"
done
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "r02_v1")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), d.get("ms_per_step_events"), "%.3e" % d["value"], d["unit"])
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-400:])
PY
cat "$O/${V}_bench_gpus2_on_1gpu.log" | tail -3
head -14 "$O/${V}_combsub_kernel_stats.csv"; head -10 "$O/${V}_sins_kernel_stats.csv"
python -c "
import json;d=json.load(open('$O/${V}_combsub_hbm_traffic.json'));print(json.dumps(d.get('__step__'),indent=1))"
