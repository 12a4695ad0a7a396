#!/bin/bash
# round 3, call D: what kind of box is this (prev build's step time), in-step per-kernel durations in one-stream order for
# prev and cur, and the per-workgroup timelines of the round-2 and the current filter kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03d}
B="python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --steps 60"
for tag in ${TAGS:-prev cur}; do
  if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
  env $lib timeout 300 $B 2>&1 | tail -1 > "$O/${V}_bench_${tag}.json"
  env $lib DDSP_HIP_ONE_STREAM=1 timeout 300 $B 2>&1 | tail -1 > "$O/${V}_bench_${tag}_one.json"
done
cd /tmp
for tag in ${TAGS:-prev cur}; do
  if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
  rm -rf "$O/prof_d"
  env $lib DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_d" -o ab -- python "$R/bench.py" --only-steps --steps 20 --warmup 3 > "$O/prof_d.log" 2>&1
  python "$R/tools/rocpd_stats.py" $(find "$O/prof_d" -name "*.db" | head -1) 2>&1 | head -10 > "$O/${V}_${tag}_one_stream_kernel_stats.csv"
  rm -rf "$O/prof_d"
done
for tag in ${TAGS:-prev cur}; do
  if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
  for mode in two one; do
    rm -rf "$O/gp"
    if [ $mode = one ]; then os=DDSP_HIP_ONE_STREAM=1; else os=Y=1; fi
    env $lib $os timeout 300 rocprofv3 --kernel-trace -d "$O/gp" -o g -- python "$R/bench.py" --only-steps --steps 12 --warmup 3 > "$O/gp.log" 2>&1
    python "$R/tools/rocpd_gaps.py" "$(find "$O/gp" -name "*.db" | head -1)" > "$O/${V}_gaps_${tag}_$mode.txt" 2>&1
    rm -rf "$O/gp"
  done
done
cd "$R"
DDSP_TL_LIB=$R/tools/ab/libddsp_hip_prevtl.so timeout 120 python tools/fir_blk_timeline.py > "$O/${V}_timeline_prev.txt" 2>&1
timeout 120 python tools/fir_blk_timeline.py > "$O/${V}_timeline_cur.txt" 2>&1
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "r03d")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), "%.3e" % d["value"], "kernel_ms", round(d["roofline"]["avg_ms"], 4))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
for f in sorted(glob.glob("gpurun_out/%s_*_kernel_stats.csv" % V)):
    print(f); print(open(f).read())
PY
for t in prev cur; do for m in two one; do echo "== step timeline $t $m"; cat "$O/${V}_gaps_${t}_$m.txt"; done; done
for t in prev cur; do echo "== timeline $t"; grep -E "launch|prologue|one pair|pair  [0-3]:|pair 1[0-3]:|end " "$O/${V}_timeline_$t.txt"; done
