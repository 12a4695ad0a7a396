#!/bin/bash
# round 3, call F: GPU test subset, then same-box A/B of CombSub and Sins steps (prev = round-2 library), one-stream in-step kernel traces
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03f}
timeout 600 python -m pytest tests/test_parity.py tests/test_fuzz.py tests/test_backward_fir.py tests/test_noise_rng.py tests/test_fullsize_gpu.py tests/test_core_api.py tests/test_baseline_shapes.py tests/test_modules.py tests/test_sharding.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$O/${V}_pytest_subset.log"
B="python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --steps 60"
for m in combsub sins; do
  for tag in prev cur; do
    if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
    env $lib timeout 300 $B --model $m 2>&1 | tail -1 > "$O/${V}_bench_${m}_${tag}.json"
  done
done
cd /tmp
for m in combsub sins; do
  for tag in prev cur; do
    if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
    rm -rf "$O/gp"
    env $lib DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace -d "$O/gp" -o g -- python "$R/bench.py" --model $m --only-steps --steps 12 --warmup 3 > "$O/gp.log" 2>&1
    python "$R/tools/rocpd_gaps.py" "$(find "$O/gp" -name "*.db" | head -1)" > "$O/${V}_gaps_${m}_${tag}_one.txt" 2>&1
    rm -rf "$O/gp"
  done
done
cd "$R"
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "r03f")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), "%.3e" % d["value"], "kernel_ms", round(d["roofline"]["avg_ms"], 4))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
PY
for m in combsub sins; do for t in prev cur; do echo "== $m $t one stream"; cat "$O/${V}_gaps_${m}_${t}_one.txt"; done; done
