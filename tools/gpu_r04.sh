#!/bin/bash
# Round 4's GPU experiments, ONE parameterised script (a gpurun call runs one or more modes):
#     gpurun -- 'V=r04_v1 bash tools/gpu_r04.sh knob DDSP_HIP_BLK_WPS "2 3"'
# modes
#   knob NAME "V1 V2 .."   same-box A/B of a launcher knob: pytest subset under every value, CombSub step (one / two streams),
#                          one-stream kernel trace, SQ counters of the filter kernel
#   libs "tagA tagB .."    same-box A/B of builds tools/ab/libddsp_hip_<tag>.so ("cur" = the in-tree library)
#   kernel "tagA tagB .."  one bench run per tag: the step and the dominant kernel alone (ablation builds)
#   gaps "tagA tagB .."    per-launch timeline of one steady-state step for every tag
#   refresh                the round's final evidence in one call (suite, smoke, every bench row, traces, counters, training, latency)
#   reference              (under tools/with_reference.sh) reference-class tests on the GPU + bench with the reference as CPU baseline
#   tests                  the whole GPU suite + smoke()
#   bench                  the driver's command (default bench.py) + kernel trace of the step
#   model NAME             bench.py --model NAME, short form
# Every output goes to gpurun_out/${V}_*; copy what is quoted into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
export V=${V:-r04}
MODEL=${MODEL:-combsub}
BENCH="python bench.py --model $MODEL --no-cpu-baseline --no-module-mode --no-live-traffic --no-also"
FIR_TESTS="tests/test_parity.py tests/test_fullsize_gpu.py tests/test_noise_rng.py tests/test_baseline_shapes.py tests/test_fuzz.py"

summary() {
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "r04")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(os.path.basename(f), "ms", round(d["ms_per_step"], 4), "events", round(d.get("ms_per_step_events") or 0, 4), "%.3e" % d["value"],
              "kernel_ms", round(r.get("avg_ms") or 0, 4), "frac", round(r.get("frac") or 0, 4))
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
PY
}

trace() {  # trace <tag> [env...]: one-stream kernel trace of the step
  tag=$1; shift
  ( cd /tmp; rm -rf "$O/prof_$tag"
    env "$@" DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_$tag" -o t -- python "$R/bench.py" --model $MODEL --only-steps --steps 20 --warmup 3 > "$O/${V}_trace_$tag.log" 2>&1
    python "$R/tools/rocpd_stats.py" $(find "$O/prof_$tag" -name "*.db" | head -1) 2>&1 | head -14 > "$O/${V}_${tag}_kernel_stats.csv"
    rm -rf "$O/prof_$tag" )
  cat "$O/${V}_${tag}_kernel_stats.csv"
}

pmc() {  # pmc <tag> <kernel-substring> [env...]: SQ counters of the step's kernels, two passes
  tag=$1; pat=$2; shift; shift
  : > "$O/${V}_pmc_$tag.txt"
  ( cd /tmp
    for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
      rm -rf "$O/spmc"
      env "$@" DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --pmc $set --kernel-trace -d "$O/spmc" -o p -- python "$R/bench.py" --model $MODEL --only-steps --steps 3 --warmup 1 > "$O/${V}_pmc_$tag.log" 2>&1
      f=$(find "$O/spmc" -name "*.db" | head -1)
      [ -n "$f" ] && python "$R/tools/rocpd_pmc.py" "$f" 2>/dev/null | grep "$pat" >> "$O/${V}_pmc_$tag.txt"
      rm -rf "$O/spmc"
    done )
  cat "$O/${V}_pmc_$tag.txt"
}

mode=${1:-tests}
case $mode in
knob)
  KNOB=$2; VALUES=$3
  for v in $VALUES; do
    echo "== pytest under $KNOB=$v"
    env $KNOB=$v timeout 900 python -m pytest $FIR_TESTS -m gpu -x -q 2>&1 | tail -3 | tee "$O/${V}_pytest_$v.log"
  done
  for rep in 1 2; do
    for v in $VALUES; do
      env $KNOB=$v timeout 300 $BENCH 2>&1 | tail -1 > "$O/${V}_bench_${v}_two_$rep.json"
      env $KNOB=$v DDSP_HIP_ONE_STREAM=1 timeout 300 $BENCH 2>&1 | tail -1 > "$O/${V}_bench_${v}_one_$rep.json"
    done
  done
  summary
  for v in $VALUES; do echo "== trace $KNOB=$v"; trace knob$v $KNOB=$v; done
  for v in $VALUES; do echo "== pmc $KNOB=$v"; pmc knob$v "${PMC_PAT:-k_fir_blk}" $KNOB=$v; done
  ;;
libs)
  # a tag is <build>[:ENV=VAL[,ENV=VAL..]]: build "cur" = the in-tree library, else tools/ab/libddsp_hip_<build>.so
  TAGS=$2
  lib_of() { b=${1%%:*}; if [ "$b" = cur ]; then echo "$R/ddsp_svc_amd/lib/libddsp_hip.so"; else echo "$R/tools/ab/libddsp_hip_$b.so"; fi; }
  env_of() { case $1 in *:*) echo "${1#*:}" | tr ',' ' ';; *) echo "X=1";; esac; }
  name_of() { echo "$1" | tr ':=,' '___'; }
  if [ -z "${NO_TESTS:-}" ]; then for t in $TAGS; do
    echo "== pytest with $t"
    env DDSP_HIP_LIB=$(lib_of $t) $(env_of $t) timeout 900 python -m pytest ${TESTS:-$FIR_TESTS} -m gpu -x -q 2>&1 | tail -3 | tee "$O/${V}_pytest_$(name_of $t).log"
  done; fi
  for rep in 1 2; do
    for t in $TAGS; do
      env DDSP_HIP_LIB=$(lib_of $t) $(env_of $t) timeout 300 $BENCH 2>&1 | tail -1 > "$O/${V}_bench_$(name_of $t)_two_$rep.json"
      env DDSP_HIP_LIB=$(lib_of $t) $(env_of $t) DDSP_HIP_ONE_STREAM=1 timeout 300 $BENCH 2>&1 | tail -1 > "$O/${V}_bench_$(name_of $t)_one_$rep.json"
    done
  done
  summary
  if [ -z "${NO_TRACE:-}" ]; then for t in $TAGS; do echo "== trace $t"; trace $(name_of $t) DDSP_HIP_LIB=$(lib_of $t) $(env_of $t); done; fi
  if [ -n "${PMC_PAT:-}" ]; then for t in $TAGS; do echo "== pmc $t"; pmc $(name_of $t) "$PMC_PAT" DDSP_HIP_LIB=$(lib_of $t) $(env_of $t); done; fi
  ;;
kernel)
  # the dominant kernel alone (bench's roofline leg) for every tag of `libs`; the step beside it once
  TAGS=$2
  lib_of() { b=${1%%:*}; if [ "$b" = cur ]; then echo "$R/ddsp_svc_amd/lib/libddsp_hip.so"; else echo "$R/tools/ab/libddsp_hip_$b.so"; fi; }
  env_of() { case $1 in *:*) echo "${1#*:}" | tr ',' ' ';; *) echo "X=1";; esac; }
  name_of() { echo "$1" | tr ':=,' '___'; }
  for t in $TAGS; do
    env DDSP_HIP_LIB=$(lib_of $t) $(env_of $t) timeout 300 $BENCH --steps 50 2>&1 | tail -1 > "$O/${V}_bench_$(name_of $t)_two_1.json"
  done
  summary
  ;;
gaps)
  # per-launch timeline of one steady-state step (two streams, then one) for every tag of `libs`
  TAGS=$2
  lib_of() { b=${1%%:*}; if [ "$b" = cur ]; then echo "$R/ddsp_svc_amd/lib/libddsp_hip.so"; else echo "$R/tools/ab/libddsp_hip_$b.so"; fi; }
  env_of() { case $1 in *:*) echo "${1#*:}" | tr ',' ' ';; *) echo "X=1";; esac; }
  name_of() { echo "$1" | tr ':=,' '___'; }
  for t in $TAGS; do for mode in two one; do
    ( cd /tmp; rm -rf "$O/gp"
      if [ $mode = one ]; then export DDSP_HIP_ONE_STREAM=1; fi
      env DDSP_HIP_LIB=$(lib_of $t) $(env_of $t) timeout 300 rocprofv3 --kernel-trace -d "$O/gp" -o g -- python "$R/bench.py" --model $MODEL --only-steps --steps 12 --warmup 3 > "$O/${V}_gp.log" 2>&1
      f=$(find "$O/gp" -name "*.db" | head -1)
      echo "== $MODEL, $t, $mode stream(s)" | tee "$O/${V}_gaps_$(name_of $t)_$mode.txt"
      python "$R/tools/rocpd_gaps.py" "$f" 2>&1 | tee -a "$O/${V}_gaps_$(name_of $t)_$mode.txt"
      rm -rf "$O/gp" )
  done; done
  ;;
refresh)
  # the round's final evidence on one box: GPU suite + smoke, the driver's command (with the reference as CPU baseline when
  # DDSP_REFERENCE_PATH is set), the other bench rows, one-stream kernel traces, SQ counters of the CombSub step, the
  # training steps, the streaming-shape latencies, cfg 4's per-GPU shape on a 1-rank communicator
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee "$O/${V}_pytest_gpu.log"
  timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v Warning | tail -4 | tee "$O/${V}_smoke.log"
  ( time timeout 600 python bench.py ) 2>"$O/${V}_bench_combsub.err" | tail -1 > "$O/${V}_bench_combsub.json"; tail -4 "$O/${V}_bench_combsub.err"
  for m in sins combsubsuperfast combsubfast rssloss mel sinesrc; do
    timeout 300 python bench.py --model $m --no-also --no-cpu-baseline --no-live-traffic 2>/dev/null | tail -1 > "$O/${V}_bench_$m.json"
  done
  timeout 300 python bench.py --cfg4 --no-also --no-cpu-baseline --no-live-traffic --no-module-mode 2>/dev/null | tail -1 > "$O/${V}_bench_cfg4_1rank.json"
  summary
  for m in combsub sins combsubsuperfast; do MODEL=$m; echo "== trace $m"; trace $m X=1 | head -8; done
  MODEL=combsub
  pmc combsub "ddsp::" X=1 > /dev/null; head -40 "$O/${V}_pmc_combsub.txt"
  for k in combsub sins combsubsuperfast combsubfast; do timeout 120 python tools/train_step_probe.py $k 2>&1 | tail -1; done | tee "$O/${V}_train_ms.txt"
  for k in combsub sins combsubsuperfast; do timeout 120 python tools/train_step_probe.py $k loss 2>&1 | tail -1; done | tee -a "$O/${V}_train_ms.txt"
  timeout 300 python tools/latency_probe.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee "$O/${V}_latency_small_shapes.txt"
  python - <<'PY'
import json, os
V = os.environ["V"]
d = json.loads(open("gpurun_out/%s_bench_combsub.json" % V).read())
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_events")})
print("roofline", {k: d["roofline"][k] for k in ("kernel", "frac", "avg_ms", "traffic")})
print("step traffic", (d.get("roofline_step_traffic") or {}).get("ratio"))
print("also", {k: round(v["ms_per_step"], 4) for k, v in d.get("also", {}).items()})
print("cpu_baseline", {k: d["cpu_baseline"].get(k) for k in ("kind", "value", "cores", "sample")})
c = json.loads(open("gpurun_out/%s_bench_cfg4_1rank.json" % V).read()).get("cfg4")
print("cfg4 (1 rank)", c and {k: c.get(k) for k in ("ms_per_step", "ms_per_step_with_gather", "value")})
PY
  ;;
reference)
  # with a reference checkout beside the snapshot (tools/with_reference.sh): the tests that hold the drop-in modules against the
  # reference's OWN classes, on the MI355X; then the bench line with cpu_baseline.kind = "reference" and the parity gate
  echo "DDSP_REFERENCE_PATH=$DDSP_REFERENCE_PATH"; ls "$DDSP_REFERENCE_PATH" | head -3
  timeout 1200 python -m pytest tests/test_modules.py tests/test_backward_fir.py tests/test_backward_fast.py tests/test_mel.py tests/test_sine_source.py tests/test_cascade_seam.py tests/test_loss.py -m gpu -q -rA -s 2>&1 | grep -v "^\s*$" | grep -E "passed|failed|PASSED|FAILED|SKIPPED|rms error|training step|Error|error" | tee "$O/${V}_reference_on_gpu.log" | tail -60
  timeout 900 python bench.py --no-module-mode --no-live-traffic --no-also 2>"$O/${V}_bench_reference.err" | tail -1 > "$O/${V}_bench_reference.json"
  timeout 900 python bench.py --model cascade_ref --batch-per-gpu 64 2>"$O/${V}_bench_cascade_ref.err" | tail -1 > "$O/${V}_bench_cascade_ref.json"
  cat "$O/${V}_bench_cascade_ref.json" | cut -c1-1500; tail -3 "$O/${V}_bench_cascade_ref.err"
  python - <<'PY'
import json, os
V = os.environ["V"]
d = json.loads(open("gpurun_out/%s_bench_reference.json" % V).read())
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "cpu_baseline", "cpu_baseline_port") if k in d}, indent=1)[:3000])
PY
  ;;
tests)
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee "$O/${V}_pytest_gpu.log"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -8 | tee "$O/${V}_smoke.log"
  ;;
bench)
  timeout 900 python bench.py 2>&1 | tail -1 > "$O/${V}_bench_default.json"
  V=$V summary
  trace default X=1
  ;;
model)
  MODEL=$2
  timeout 600 python bench.py --model $MODEL --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>&1 | tail -1 > "$O/${V}_bench_$MODEL.json"
  summary
  trace $MODEL X=1
  ;;
*)
  echo "unknown mode $mode"; exit 2;;
esac
