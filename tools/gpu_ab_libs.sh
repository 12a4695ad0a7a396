#!/bin/bash
# same-box A/B of several builds of the library (tools/ab/libddsp_hip_<tag>.so, TAGS="..."; "cur" = the in-tree one)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-abl}
for rep in 1 2; do
  for tag in ${TAGS:-cur}; do
    if [ $tag = cur ]; then lib=""; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
    env $lib timeout 300 python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_${tag}_$rep.json"
  done
done
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "abl")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), "%.3e" % d["value"], "kernel_ms", round(d["roofline"]["avg_ms"], 4))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
PY
