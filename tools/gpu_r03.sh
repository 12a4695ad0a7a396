#!/bin/bash
# Round-3 refresh in one gpurun call: GPU suite, smoke, the driver's own bench command (headline + also + live traffic + CPU
# baselines), Sins, cfg 4's per-GPU shape on a 1-rank RCCL communicator, cfg-5 seam, --gpus 2 refusal, the other models,
# per-kernel traces in one-stream order, SQ counters of the steps, training-step traces.  Every profiler run is bounded.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03_v31}
echo "== host =="; nproc; lscpu | grep -m1 "Model name"
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$O/${V}_pytest_gpu.log"
  timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -4 | tee "$O/${V}_smoke.log"
fi
( time timeout 400 python bench.py ) 2>&1 | tail -5 > "$O/${V}_bench_combsub.log"; grep '^{' "$O/${V}_bench_combsub.log" | tail -1 > "$O/${V}_bench_combsub.json"
timeout 300 python bench.py --model sins --no-also 2>&1 | tail -1 > "$O/${V}_bench_sins.json"
timeout 200 python bench.py --cfg4 --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --steps 30 2>&1 | tail -1 > "$O/${V}_bench_cfg4_1rank.json"
timeout 100 python bench.py --gpus 2 --no-cpu-baseline > "$O/${V}_bench_gpus2_on_1gpu.log" 2>&1; echo "exit $?" >> "$O/${V}_bench_gpus2_on_1gpu.log"
timeout 300 python bench.py --model cascade_seam --batch-per-gpu 64 2>&1 | tail -1 > "$O/${V}_bench_cascade_seam.json"
DDSP_HIP_ONE_STREAM=1 timeout 200 python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_combsub_one_stream.json"
for m in combsubfast combsubsuperfast mel sinesrc rssloss; do
  timeout 200 python bench.py --model $m --no-cpu-baseline 2>&1 | tail -1 > "$O/${V}_bench_$m.json"
done
cd /tmp
for m in combsub sins combsubsuperfast; do
  rm -rf "$O/prof_$m"
  DDSP_HIP_ONE_STREAM=1 timeout 120 rocprofv3 --kernel-trace -d "$O/prof_$m" -o $m -- python "$R/bench.py" --model $m --only-steps --steps 20 --warmup 3 > "$O/prof_$m.log" 2>&1
  python "$R/tools/rocpd_stats.py" "$(find "$O/prof_$m" -name "*.db" | head -1)" 2>&1 | head -16 > "$O/${V}_${m}_kernel_stats.csv"
  python "$R/tools/rocpd_gaps.py" "$(find "$O/prof_$m" -name "*.db" | head -1)" > "$O/${V}_${m}_step_timeline_one_stream.txt" 2>&1
  rm -rf "$O/prof_$m"
done
cd "$R"
MODELS="combsub sins" bash tools/gpu_step_pmc.sh > /dev/null 2>&1
cp "$O/step_pmc_combsub.txt" "$O/${V}_step_pmc_combsub.txt"; cp "$O/step_pmc_sins.txt" "$O/${V}_step_pmc_sins.txt"
MODELS="combsub sins combsubsuperfast" PROF_TIMEOUT=90 PROBE_ARGS=loss V=$V bash tools/gpu_train_prof.sh > "$O/${V}_train.log" 2>&1
grep -h "forward+backward" "$O"/tp_*.log | tee "$O/${V}_train_ms.txt"
for k in combsub sins combsubsuperfast combsubfast; do timeout 120 python tools/train_step_probe.py $k 2>&1 | tail -1; done | tee -a "$O/${V}_train_ms.txt"
cd /tmp; rm -rf "$O/kp"; timeout 200 rocprofv3 --kernel-trace -d "$O/kp" -o k -- python "$R/bench.py" --model rssloss --steps 20 --warmup 3 > "$O/kp.log" 2>&1
python "$R/tools/rocpd_stats.py" $(find "$O/kp" -name "*.db" | head -1) 2>&1 | head -24 > "$O/${V}_rssloss_kernel_stats.csv"; rm -rf "$O/kp"; cd "$R"
timeout 300 python tools/race_probe.py 10 2>&1 | tail -40 > "$O/${V}_race_probe.txt"; echo "race probe: $(grep -c '0 mismatches' "$O/${V}_race_probe.txt") operations clean"; grep -v "0 mismatches" "$O/${V}_race_probe.txt" | grep -v informational | tail -3
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "r03_v31")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), d.get("ms_per_step_events"), "%.3e" % d["value"], d["unit"])
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
PY
cat "$O/${V}_bench_gpus2_on_1gpu.log" | tail -2
head -12 "$O/${V}_combsub_kernel_stats.csv"; head -9 "$O/${V}_sins_kernel_stats.csv"
