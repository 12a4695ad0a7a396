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
    CSV_OUT="$O/${V}_${tag}_steady_kernel_stats.csv" python "$R/tools/rocpd_launches.py" "$f" k_phase_frame_sums ${SERIES_POS:-4} 2>&1 > "$O/${V}_${tag}_launches.txt"
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
final)
  # the round's final evidence on ONE box (run under tools/with_reference.sh): GPU suite + smoke (reference-class tests included),
  # the driver's command (reference as CPU baseline; in-step trace table saved), the other bench rows, the B sweep, steady-state
  # traces (the last 150 of 600 steps), SQ counters of the step's kernels, training steps (with and without the reference's loss),
  # streaming-shape latencies, the step + mel pair, the communicator check, cfg 5 with the reference's networks
  timeout 2400 python -m pytest tests -m gpu -q -rs 2>&1 | grep -E "passed|failed|error|SKIPPED" | tee "$O/${V}_pytest_gpu.log"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee "$O/${V}_smoke.log"
  ( time timeout 1200 python bench.py --trace-stats-out "$O/${V}_bench_in_step_kernel_stats.csv" ) 2>"$O/${V}_bench_combsub.err" | tail -1 > "$O/${V}_bench_combsub.json"; tail -4 "$O/${V}_bench_combsub.err"
  line "$O/${V}_bench_combsub.json"
  for m in sins combsubsuperfast combsubfast rssloss mel sinesrc; do
    timeout 300 python bench.py --model $m --no-also --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 > "$O/${V}_bench_$m.json"
  done
  echo "== B sweep"; for B in 16 32 64 128 256; do f="$O/${V}_sweep_B${B}.json"; timeout 300 $BENCH --batch-per-gpu $B --steps 60 2>&1 | tail -1 > "$f"; line "$f"; done
  export LAST=150
  for m in combsub sins; do MODEL=$m; echo "== steady-state trace, $m"; TRACE_STEPS=600 SERIES_POS=3 trace $m X=1 > /dev/null; cat "$O/${V}_${m}_launches.txt" | head -12; done
  MODEL=combsub
  echo "== one launch per kernel (rounds 2 - 5's layout, knob STREAM_LAYOUT = 4, one stream): the kernels' own steady-state times"
  TRACE_STEPS=600 SERIES_POS=4 trace layout4_one_stream DDSP_HIP_STREAM_LAYOUT=4 DDSP_HIP_ONE_STREAM=1 > /dev/null; head -12 "$O/${V}_layout4_one_stream_launches.txt"; cat "$O/${V}_layout4_one_stream_steady_kernel_stats.csv"
  echo "== two-stream layout of rounds 2 - 5 (knob STREAM_LAYOUT = 4), same box"
  for rep in 1 2; do for t in "fused:X=1" "two_streams:DDSP_HIP_STREAM_LAYOUT=4"; do name=${t%%:*}; f="$O/${V}_bench_${name}_$rep.json"; env ${t#*:} timeout 300 $BENCH 2>&1 | tail -1 > "$f"; line "$f"; done; done
  MODELS=combsub bash tools/gpu_step_pmc.sh > /dev/null 2>&1; cp "$O/step_pmc_combsub.txt" "$O/${V}_pmc_combsub.txt"; grep -c . "$O/${V}_pmc_combsub.txt"
  for k in combsub sins combsubsuperfast combsubfast combsub512; do timeout 300 python tools/train_step_probe.py $k 2>&1 | tail -1; done | tee "$O/${V}_train_ms.txt"
  for k in combsub sins combsubsuperfast; do timeout 300 python tools/train_step_probe.py $k loss 2>&1 | tail -1; done | tee -a "$O/${V}_train_ms.txt"
  ( cd /tmp; rm -rf "$O/tp"; TRAIN_WARM=300 TRAIN_STEPS=100 timeout 300 rocprofv3 --kernel-trace -d "$O/tp" -o t -- python "$R/tools/train_step_probe.py" combsub > /dev/null 2>&1
    f=$(find "$O/tp" -name "*.db" | head -1)
    LAST=100 python "$R/tools/rocpd_launches.py" "$f" k_phase_frame_sums 2>&1 | head -40 > "$O/${V}_train_combsub_launches.txt"; rm -rf "$O/tp" )
  timeout 300 python tools/latency_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee "$O/${V}_latency_small_shapes.txt"
  timeout 300 python tools/mel_pair_probe.py 2>&1 | tail -1 | tee "$O/${V}_mel_pair.txt"
  timeout 300 python tools/mel_shifted_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee "$O/${V}_mel_shifted.txt"
  timeout 300 python bench.py --model mel --keyshift 3 --no-also --no-live-traffic 2>/dev/null | tail -1 > "$O/${V}_bench_mel_keyshift3.json"; cut -c1-400 "$O/${V}_bench_mel_keyshift3.json"
  echo "== communicator" | tee "$O/${V}_pg_check.txt"
  for q in 4 8; do for m in none nccl; do GPU_MAX_HW_QUEUES=$q timeout 200 python tools/pg_probe.py $m 2>&1 | grep "ms/step" | sed "s/^/queues=$q /" | tee -a "$O/${V}_pg_check.txt"; done; done
  if [ -n "${DDSP_REFERENCE_PATH:-}" ]; then
    timeout 1200 python bench.py --model cascade_ref --batch-per-gpu 64 2>"$O/${V}_bench_cascade_ref.err" | tail -1 > "$O/${V}_bench_cascade_ref.json"
    cut -c1-900 "$O/${V}_bench_cascade_ref.json"; tail -2 "$O/${V}_bench_cascade_ref.err"
  fi
  python - <<'PY'
import json, os
V = os.environ["V"]
d = json.loads(open("gpurun_out/%s_bench_combsub.json" % V).read())
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_events")})
print("parity", d.get("parity_vs_oracle"))
r = d["roofline"]; print("roofline", {k: r[k] for k in r if k not in ("note", "traffic_detail", "in_step_kernels")})
print("step traffic", (d.get("roofline_step_traffic") or {}).get("ratio"))
c = d.get("cfg4"); print("cfg4", c and {k: c.get(k) for k in ("ms_per_step", "ms_per_step_with_gather", "ms_per_step_with_gather_async", "value")})
print("also", {k: round(v["ms_per_step"], 4) for k, v in d.get("also", {}).items()})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("kind", "value", "cores", "sample", "gpu_over_cpu")})
print("module mode", d.get("value_module_mode"))
for m in ("sins", "combsubsuperfast", "combsubfast", "rssloss", "mel", "sinesrc"):
    try:
        e = json.loads(open("gpurun_out/%s_bench_%s.json" % (V, m)).read())
        print(m, round(e["ms_per_step"], 4), e.get("in_kernel_noise", {}).get("ms_per_step"))
    except Exception as ex:
        print(m, "ERR", ex)
PY
  ;;
default)
  ( time timeout 900 python bench.py ) 2>"$O/${V}_bench_default.err" | tail -1 > "$O/${V}_bench_default.json"; tail -4 "$O/${V}_bench_default.err"
  line "$O/${V}_bench_default.json"
  ;;
*) echo "unknown mode $mode"; exit 2;;
esac
