#!/bin/bash
# round 3, call C: same-box A/B/C of library builds (TAGS) -- step time (two reps) and the step's kernel trace per build
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03c}
TAGS=${TAGS:-"prev cur"}
for rep in 1 2; do
  for tag in $TAGS; do
    if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
    env $lib timeout 300 python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_${tag}_$rep.json"
  done
done
cd /tmp
for tag in $TAGS; do
  if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
  rm -rf "$O/prof_c"
  env $lib timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_c" -o ab -- python "$R/bench.py" --only-steps --steps 20 --warmup 3 > "$O/prof_c.log" 2>&1
  python "$R/tools/rocpd_stats.py" $(find "$O/prof_c" -name "*.db" | head -1) 2>&1 | head -12 > "$O/${V}_${tag}_kernel_stats.csv"
  rm -rf "$O/prof_c"
done
cd "$R"
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "r03c")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), "%.3e" % d["value"], "kernel_ms", round(d["roofline"]["avg_ms"], 4))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
for f in sorted(glob.glob("gpurun_out/%s_*_kernel_stats.csv" % V)):
    print(f); print(open(f).read())
PY
