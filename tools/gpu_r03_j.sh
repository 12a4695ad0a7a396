#!/bin/bash
# round 3, call J: the whole GPU suite, then prev / cur step times (two and one stream) and the two-stream step timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03j}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$O/${V}_pytest_gpu.log"
B="python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --steps 60"
for tag in prev cur; do
  if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
  env $lib timeout 200 $B 2>&1 | tail -1 > "$O/${V}_bench_${tag}.json"
  env $lib DDSP_HIP_ONE_STREAM=1 timeout 200 $B 2>&1 | tail -1 > "$O/${V}_bench_${tag}_one.json"
done
cd /tmp
rm -rf "$O/gp"; timeout 100 rocprofv3 --kernel-trace -d "$O/gp" -o g -- python "$R/bench.py" --only-steps --steps 12 --warmup 3 > "$O/gp.log" 2>&1
python "$R/tools/rocpd_gaps.py" "$(find "$O/gp" -name "*.db" | head -1)" > "$O/${V}_gaps_cur_two.txt" 2>&1; rm -rf "$O/gp"
cd "$R"
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "r03j")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), "kernel_ms", round(d["roofline"]["avg_ms"], 4))
PY
cat "$O/${V}_gaps_cur_two.txt"
