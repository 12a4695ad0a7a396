#!/bin/bash
# Round 5's GPU experiments (same conventions as gpu_r04.sh, whose modes stay available: `bash tools/gpu_r04.sh <mode> ..`):
#     gpurun -- 'V=r05_v1 bash tools/gpu_r05.sh lanes'
# modes
#   lanes        sub-batch lanes (csrc/api.hip): GPU tests, same-box A/B of LANE_ROWS / LANES at B = 32 and 64 (interleaved, 2 reps),
#                the B sweep {16,32,64,128,256} split / unsplit, a multi-stream timeline of one step, the driver's command
#   sweep        only the B sweep
#   default      the driver's command + its kernel trace
# Every output goes to gpurun_out/${V}_*; copy what is quoted into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
export V=${V:-r05}
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

ab() {  # ab <B> <tag:ENV=VAL,ENV=VAL> ...   interleaved, 2 reps
  B=$1; shift
  for rep in 1 2; do
    for t in "$@"; do
      name=${t%%:*}; envs=$(echo "${t#*:}" | tr ',' ' ')
      f="$O/${V}_bench_B${B}_${name}_$rep.json"
      env $envs timeout 300 $BENCH --batch-per-gpu $B 2>&1 | tail -1 > "$f"
      line "$f"
    done
  done
}

sweep() {
  for B in 16 32 64 128 256; do
    for t in "split:X=1" "unsplit:DDSP_HIP_LANE_ROWS=1"; do
      name=${t%%:*}; envs=${t#*:}
      f="$O/${V}_sweep_B${B}_${name}.json"
      env $envs timeout 300 $BENCH --batch-per-gpu $B --steps 60 2>&1 | tail -1 > "$f"
      line "$f"
    done
  done
}

gaps() {  # gaps <tag> [env..]: per-launch timeline of steady-state steps, all streams
  tag=$1; shift
  ( cd /tmp; rm -rf "$O/gp"
    env "$@" timeout 300 rocprofv3 --kernel-trace -d "$O/gp" -o g -- python "$R/bench.py" --model $MODEL --only-steps --steps 12 --warmup 3 ${GAPS_ARGS:-} > "$O/${V}_gp.log" 2>&1
    f=$(find "$O/gp" -name "*.db" | head -1)
    python "$R/tools/rocpd_gaps.py" "$f" 2>&1 > "$O/${V}_gaps_$tag.txt"
    python "$R/tools/rocpd_stats.py" "$f" 2>&1 | head -14 > "$O/${V}_${tag}_kernel_stats.csv"
    rm -rf "$O/gp" )
  head -60 "$O/${V}_gaps_$tag.txt"; cat "$O/${V}_${tag}_kernel_stats.csv"
}

mode=${1:-lanes}
case $mode in
lanes)
  timeout 900 python -m pytest tests/test_lanes.py tests/test_small_shapes.py tests/test_noise_rng.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -4 | tee "$O/${V}_pytest_lanes.log"
  echo "== B = 32"; ab 32 "off:DDSP_HIP_LANE_ROWS=1" "lanes16:X=1" "serial16:DDSP_HIP_LANES=1" "lanes8:DDSP_HIP_LANE_ROWS=7000"
  echo "== B = 64"; ab 64 "off:DDSP_HIP_LANE_ROWS=1" "lanes16:X=1" "lanes32:DDSP_HIP_LANE_ROWS=28000" "serial16:DDSP_HIP_LANES=1"
  echo "== sweep"; sweep
  echo "== timeline, lanes"; gaps lanes X=1 | head -80
  echo "== timeline, unsplit"; gaps unsplit DDSP_HIP_LANE_ROWS=1 | head -50
  echo "== the driver's command"
  ( time timeout 900 python bench.py ) 2>"$O/${V}_bench_default.err" | tail -1 > "$O/${V}_bench_default.json"; tail -4 "$O/${V}_bench_default.err"
  python - <<'PY'
import json, os
V = os.environ["V"]
d = json.loads(open("gpurun_out/%s_bench_default.json" % V).read())
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_events")})
print("parity", d.get("parity_vs_oracle"))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "frac", "avg_ms", "traffic")})
print("cfg4", d.get("cfg4"))
print("also", {k: round(v["ms_per_step"], 4) for k, v in d.get("also", {}).items()})
print("gpu chain", (d.get("cpu_baseline_aten_chain") or {}).get("same_chain_on_gpu"))
PY
  ;;
suite)
  # the whole GPU suite + smoke, the lanes probe (split against unsplit at full size, per output and utterance), the cfg-4
  # line with and without the allocator's cache released first, the NSF source line (both draw forms)
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee "$O/${V}_pytest_gpu.log"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee "$O/${V}_smoke.log"
  timeout 300 python tools/lanes_probe.py 14336 2>&1 | grep -v Warning | tee "$O/${V}_lanes_probe.txt"
  for t in "fresh:" "keep:--cfg4-keep-cache" "fresh20:--cfg4-round-steps 20"; do
    name=${t%%:*}; args=${t#*:}
    timeout 600 python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --cfg4 $args 2>/dev/null | tail -1 > "$O/${V}_bench_cfg4_$name.json"
    python - "$O/${V}_bench_cfg4_$name.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read()); c = d["cfg4"]
print(sys.argv[1].split("/")[-1], "headline", round(d["ms_per_step"], 4), "cfg4", round(c["ms_per_step"], 4), "with gather", round(c.get("ms_per_step_with_gather", 0), 4),
      "rounds", c["rounds_ms"], c.get("rounds_ms_with_gather"))
PY
  done
  timeout 300 python bench.py --model sinesrc 2>/dev/null | tail -1 > "$O/${V}_bench_sinesrc.json"
  python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/%s_bench_sinesrc.json" % os.environ["V"]).read())
print("sinesrc randn+kernel ms", round(d["ms_per_step"], 4), "peak MB", d["peak_bytes_per_call"] / 1e6, "| in-kernel ms", round(d["in_kernel_noise"]["ms_per_step"], 4),
      "peak MB", d["in_kernel_noise"]["peak_bytes_per_call"] / 1e6, "| kernel only", round(d["kernel_only"]["ms_per_step"], 4), d["in_kernel_noise"]["draw_moments"])
PY
  ;;
probe3)
  # the GPU suite; why a process with an RCCL communicator runs the B = 32 step 10 % slower (tools/pg_probe.py); the long-tap
  # adjoints, FFT form against the direct correlations (knob FIR_BWD_DIRECT), and a training step at 256 / 512 / 256 bins
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee "$O/${V}_pytest_gpu.log"
  : > "$O/${V}_pg_probe.txt"
  for m in none nccl nccl_nomon gloo nccl_destroyed none; do
    timeout 200 python tools/pg_probe.py $m 2>&1 | grep "ms/step" | tee -a "$O/${V}_pg_probe.txt"
  done
  for m in none nccl; do timeout 200 python tools/pg_probe.py $m graph 2>&1 | grep "ms/step" | tee -a "$O/${V}_pg_probe.txt"; done
  for m in none nccl; do PROBE_B=16 timeout 200 python tools/pg_probe.py $m 2>&1 | grep "ms/step" | tee -a "$O/${V}_pg_probe.txt"; done
  for k in 0 1; do
    echo "FIR_BWD_DIRECT=$k" | tee -a "$O/${V}_fir_bwd_long.txt"
    NBINS=512 DDSP_HIP_FIR_BWD_DIRECT=$k timeout 300 python tools/fir_bwd_bench.py 2>&1 | tail -1 | tee -a "$O/${V}_fir_bwd_long.txt"
    NBINS=300 DDSP_HIP_FIR_BWD_DIRECT=$k timeout 300 python tools/fir_bwd_bench.py 2>&1 | tail -1 | tee -a "$O/${V}_fir_bwd_long.txt"
  done
  ;;
pg)
  # the RCCL communicator's 32 us per step: which part of it?  (one configuration per process, tools/pg_probe.py)
  : > "$O/${V}_pg_probe.txt"
  run() { echo "-- $*" | tee -a "$O/${V}_pg_probe.txt"; env "$@" 2>&1 | grep "ms/step" | tee -a "$O/${V}_pg_probe.txt"; }
  run X=1 timeout 200 python tools/pg_probe.py none
  run X=1 timeout 200 python tools/pg_probe.py nccl
  run DDSP_HIP_ONE_STREAM=1 timeout 200 python tools/pg_probe.py none
  run DDSP_HIP_ONE_STREAM=1 timeout 200 python tools/pg_probe.py nccl
  run X=1 timeout 200 python tools/pg_probe.py streams8
  run X=1 timeout 200 python tools/pg_probe.py streams32
  run X=1 timeout 200 python tools/pg_probe.py pinned
  run GPU_MAX_HW_QUEUES=8 timeout 200 python tools/pg_probe.py nccl
  run GPU_MAX_HW_QUEUES=2 timeout 200 python tools/pg_probe.py nccl
  run GPU_MAX_HW_QUEUES=2 timeout 200 python tools/pg_probe.py none
  run NCCL_MAX_NCHANNELS=1 NCCL_MIN_NCHANNELS=1 timeout 200 python tools/pg_probe.py nccl
  run HSA_ENABLE_SDMA=0 timeout 200 python tools/pg_probe.py nccl
  run RCCL_MSCCL_ENABLE=0 RCCL_MSCCLPP_ENABLE=0 timeout 200 python tools/pg_probe.py nccl
  run NCCL_LAUNCH_MODE=GROUP timeout 200 python tools/pg_probe.py nccl
  run HIP_FORCE_DEV_KERNARG=1 timeout 200 python tools/pg_probe.py nccl
  run NCCL_DEBUG=INFO timeout 200 python tools/pg_probe.py nccl
  NCCL_DEBUG=INFO timeout 200 python tools/pg_probe.py nccl 2>&1 | grep -i "NCCL INFO" | head -60 > "$O/${V}_nccl_info.txt"
  ;;
pg2)
  # the fix: the second stream on a hardware queue of its own -- by priority class (DDSP_HIP_AUX_PRIORITY=-1) or by more queues
  : > "$O/${V}_pg_fix.txt"
  run() { echo "-- $*" | tee -a "$O/${V}_pg_fix.txt"; env "$@" 2>&1 | grep "ms/step" | tee -a "$O/${V}_pg_fix.txt"; }
  for rep in 1 2; do
    run X=1 timeout 200 python tools/pg_probe.py none
    run DDSP_HIP_AUX_PRIORITY=-1 timeout 200 python tools/pg_probe.py none
    run X=1 timeout 200 python tools/pg_probe.py nccl
    run DDSP_HIP_AUX_PRIORITY=-1 timeout 200 python tools/pg_probe.py nccl
    run GPU_MAX_HW_QUEUES=8 timeout 200 python tools/pg_probe.py none
    run GPU_MAX_HW_QUEUES=8 timeout 200 python tools/pg_probe.py nccl
  done
  run GPU_MAX_HW_QUEUES=16 timeout 200 python tools/pg_probe.py nccl
  run GPU_MAX_HW_QUEUES=8 PROBE_B=16 timeout 200 python tools/pg_probe.py nccl
  run DDSP_HIP_AUX_PRIORITY=-1 PROBE_B=16 timeout 200 python tools/pg_probe.py nccl
  run X=1 PROBE_B=16 timeout 200 python tools/pg_probe.py none
  ;;
final)
  # the round's final evidence on one box (run under tools/with_reference.sh): round 4's refresh (suite + smoke, the driver's
  # command with the reference as CPU baseline, the other bench rows, traces, counters, training steps, latencies) plus round 5's
  # rows: the B sweep, the NSF source with both draw forms, the classic configuration's training step with the FFT-form
  # long-tap adjoint and with the direct correlations, the communicator check
  V=$V bash tools/gpu_r04.sh refresh
  echo "== B sweep"; for B in 16 32 64 128 256; do
    f="$O/${V}_sweep_B${B}.json"; timeout 300 $BENCH --batch-per-gpu $B --steps 60 2>&1 | tail -1 > "$f"; line "$f"; done
  timeout 300 python bench.py --model sinesrc 2>/dev/null | tail -1 > "$O/${V}_bench_sinesrc.json"
  echo "== training at 256 / 512 / 256 bins" | tee -a "$O/${V}_train_ms.txt"
  for k in 0 1; do echo "FIR_BWD_DIRECT=$k" | tee -a "$O/${V}_train_ms.txt"
    DDSP_HIP_FIR_BWD_DIRECT=$k timeout 200 python tools/train_step_probe.py combsub512 2>&1 | tail -1 | tee -a "$O/${V}_train_ms.txt"; done
  for k in 0 1; do echo "FIR_BWD_DIRECT=$k" | tee -a "$O/${V}_fir_bwd_long.txt"
    NBINS=512 DDSP_HIP_FIR_BWD_DIRECT=$k timeout 300 python tools/fir_bwd_bench.py 2>&1 | tail -1 | tee -a "$O/${V}_fir_bwd_long.txt"; done
  if [ -f tools/ab/libddsp_hip_ffb3.so ]; then echo "three workgroups per CU (tools/ab/libddsp_hip_ffb3.so)" | tee -a "$O/${V}_fir_bwd_long.txt"
    NBINS=512 DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_ffb3.so timeout 300 python tools/fir_bwd_bench.py 2>&1 | tail -1 | tee -a "$O/${V}_fir_bwd_long.txt"; fi
  echo "== communicator" | tee "$O/${V}_pg_check.txt"
  for q in 4 8; do for m in none nccl; do GPU_MAX_HW_QUEUES=$q timeout 200 python tools/pg_probe.py $m 2>&1 | grep "ms/step" | sed "s/^/queues=$q /" | tee -a "$O/${V}_pg_check.txt"; done; done
  ;;
check)
  # the final tree once more: GPU suite (with the reference when DDSP_REFERENCE_PATH is set) + smoke, the long-tap adjoint as shipped,
  # the classic configuration's training step, the training steps
  timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee "$O/${V}_pytest_gpu.log"
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5 | tee "$O/${V}_smoke.log"
  for k in 0 1; do echo "FIR_BWD_DIRECT=$k" | tee -a "$O/${V}_fir_bwd_long.txt"
    NBINS=512 DDSP_HIP_FIR_BWD_DIRECT=$k timeout 300 python tools/fir_bwd_bench.py 2>&1 | tail -1 | tee -a "$O/${V}_fir_bwd_long.txt"; done
  for k in combsub sins combsub512; do timeout 200 python tools/train_step_probe.py $k 2>&1 | tail -1; done | tee "$O/${V}_train_ms.txt"
  ( cd /tmp; rm -rf "$O/tp"; timeout 300 rocprofv3 --kernel-trace --stats -d "$O/tp" -o t -- python "$R/tools/train_step_probe.py" combsub512 > /dev/null 2>&1
    python "$R/tools/rocpd_stats.py" $(find "$O/tp" -name "*.db" | head -1) 2>&1 | head -16 > "$O/${V}_train_combsub512_kernel_stats.csv"; rm -rf "$O/tp" )
  cat "$O/${V}_train_combsub512_kernel_stats.csv"
  ;;
layouts)
  # stream layouts 5 - 7 (the harmonic taps synthesised early on the branch stream, a third tap buffer) against the default 4;
  # needs profiles/r05_v9_stream_layouts_5_7.patch applied to csrc/api.hip (measured, lost, not in the tree: EXPERIMENTS 5.9)
  echo "== B = 32"; ab 32 "L4:X=1" "L5:DDSP_HIP_STREAM_LAYOUT=5" "L6:DDSP_HIP_STREAM_LAYOUT=6" "L7:DDSP_HIP_STREAM_LAYOUT=7"
  echo "== B = 64"; ab 64 "L4:X=1" "L5:DDSP_HIP_STREAM_LAYOUT=5" "L7:DDSP_HIP_STREAM_LAYOUT=7"
  ;;
bwdpmc)
  # the long-tap adjoint's own evidence: kernel trace, HBM bytes (FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only;
  # FETCH_SIZE doubled per the gfx950 correction), SQ counters
  ( cd /tmp
    rm -rf "$O/bp"; NBINS=512 timeout 300 rocprofv3 --kernel-trace --stats -d "$O/bp" -o t -- python "$R/tools/fir_bwd_bench.py" > /dev/null 2>&1
    python "$R/tools/rocpd_stats.py" $(find "$O/bp" -name "*.db" | head -1) 2>&1 | head -6 | tee "$O/${V}_fir_fft_bwd_kernel_stats.csv"; rm -rf "$O/bp"
    : > "$O/${V}_fir_fft_bwd_pmc.txt"
    for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
               "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
      rm -rf "$O/bp"; NBINS=512 timeout 300 rocprofv3 --pmc $set --kernel-trace -d "$O/bp" -o p -- python "$R/tools/fir_bwd_bench.py" > /dev/null 2>&1
      f=$(find "$O/bp" -name "*.db" | head -1)
      [ -n "$f" ] && python "$R/tools/rocpd_pmc.py" "$f" 2>/dev/null | grep "k_fir_fft_bwd" >> "$O/${V}_fir_fft_bwd_pmc.txt"
      rm -rf "$O/bp"
    done )
  cat "$O/${V}_fir_fft_bwd_pmc.txt"
  ;;
sweep) sweep ;;
default)
  ( time timeout 900 python bench.py ) 2>"$O/${V}_bench_default.err" | tail -1 > "$O/${V}_bench_default.json"; tail -4 "$O/${V}_bench_default.err"
  line "$O/${V}_bench_default.json"
  ;;
*) echo "unknown mode $mode"; exit 2;;
esac
