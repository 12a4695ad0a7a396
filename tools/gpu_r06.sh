#!/bin/bash
# Round 6's GPU experiments (conventions of gpu_r05.sh, whose modes stay available):
#     gpurun -- 'V=r06_v1 bash tools/gpu_r06.sh base'
# modes
#   base         the quick bench line, a two-stream timeline of one step, the one-stream kernel trace (per-kernel in-step times)
#   ab           same-box A/B of library builds: LIBS="tag:path tag:path" (DDSP_HIP_LIB), interleaved, REPS reps, + one-stream traces
#   knobs        same-box A/B of launcher knobs: TAGS="tag:ENV=VAL,ENV=VAL .."
#   default      the driver's command
# Every output goes to gpurun_out/${V}_*; copy what is quoted into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
export V=${V:-r06}
MODEL=${MODEL:-combsub}
BENCH="python bench.py --model $MODEL --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --no-cfg4"

line() {  # line <file>: ms_per_step, events, kernel
python - "$1" <<'PY'
import json, sys, os
f = sys.argv[1]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print("%-52s ms %.4f events %.4f  %.3e samples/s  kernel_ms %.4f" % (os.path.basename(f), d["ms_per_step"], d.get("ms_per_step_events") or 0,
          d["value"], r.get("avg_ms") or 0))
except Exception as e:
    print(f, "ERR", e, open(f).read()[-400:])
PY
}

trace() {  # trace <tag> [env..]: kernel trace of steady-state steps -> timeline of the last step + per-kernel stats
  tag=$1; shift
  ( cd /tmp; rm -rf "$O/gp"
    env "$@" timeout 300 rocprofv3 --kernel-trace -d "$O/gp" -o g -- python "$R/bench.py" --model $MODEL --only-steps --steps ${TRACE_STEPS:-24} --warmup 3 ${TRACE_ARGS:-} > "$O/${V}_gp.log" 2>&1
    f=$(find "$O/gp" -name "*.db" | head -1)
    python "$R/tools/rocpd_gaps.py" "$f" 2>&1 > "$O/${V}_gaps_$tag.txt"
    python "$R/tools/rocpd_stats.py" "$f" 2>&1 | head -14 > "$O/${V}_${tag}_kernel_stats.csv"
    python "$R/tools/rocpd_launches.py" "$f" k_phase_frame_sums ${SERIES_POS:-4} 2>&1 > "$O/${V}_${tag}_launches.txt"
    rm -rf "$O/gp" )
  head -40 "$O/${V}_gaps_$tag.txt"; head -8 "$O/${V}_${tag}_kernel_stats.csv"; cat "$O/${V}_${tag}_launches.txt"
}

mode=${1:-base}
case $mode in
base)
  for rep in 1 2; do f="$O/${V}_bench_quick_$rep.json"; timeout 300 $BENCH 2>&1 | tail -1 > "$f"; line "$f"; done
  echo "== two streams"; trace two X=1
  echo "== one stream"; trace one DDSP_HIP_ONE_STREAM=1
  ;;
ab)
  REPS=${REPS:-2}
  for rep in $(seq 1 $REPS); do
    for t in $LIBS; do
      name=${t%%:*}; lib=${t#*:}
      f="$O/${V}_bench_${name}_$rep.json"
      DDSP_HIP_LIB="$R/$lib" timeout 300 $BENCH ${AB_ARGS:-} 2>&1 | tail -1 > "$f"
      line "$f"
    done
  done
  if [ "${AB_TRACE:-1}" = 1 ]; then
    for t in $LIBS; do name=${t%%:*}; lib=${t#*:}; echo "== one stream, $name"; trace "one_$name" DDSP_HIP_ONE_STREAM=1 DDSP_HIP_LIB="$R/$lib"; done
  fi
  ;;
knobs)
  REPS=${REPS:-2}
  for rep in $(seq 1 $REPS); do
    for t in $TAGS; do
      name=${t%%:*}; envs=$(echo "${t#*:}" | tr ',' ' ')
      f="$O/${V}_bench_${name}_$rep.json"
      env $envs timeout 300 $BENCH ${AB_ARGS:-} 2>&1 | tail -1 > "$f"
      line "$f"
    done
  done
  ;;
train)
  # forward + backward steps at the clocks' steady state, and the kernels of the CombSub training step (last 100 steps of 400)
  for k in ${KINDS:-combsub sins combsubsuperfast combsubfast}; do timeout 300 python tools/train_step_probe.py $k 2>&1 | tail -1; done | tee "$O/${V}_train_ms.txt"
  ( cd /tmp; rm -rf "$O/tp"; TRAIN_WARM=300 TRAIN_STEPS=100 timeout 300 rocprofv3 --kernel-trace -d "$O/tp" -o t -- python "$R/tools/train_step_probe.py" ${TRACE_KIND:-combsub} > /dev/null 2>&1
    f=$(find "$O/tp" -name "*.db" | head -1)
    LAST=100 python "$R/tools/rocpd_launches.py" "$f" k_phase_frame_sums 2>&1 | head -40 > "$O/${V}_train_${TRACE_KIND:-combsub}_launches.txt"; rm -rf "$O/tp" )
  cat "$O/${V}_train_${TRACE_KIND:-combsub}_launches.txt"
  ;;
default)
  ( time timeout 900 python bench.py ) 2>"$O/${V}_bench_default.err" | tail -1 > "$O/${V}_bench_default.json"; tail -4 "$O/${V}_bench_default.err"
  line "$O/${V}_bench_default.json"
  ;;
*) echo "unknown mode $mode"; exit 2;;
esac
