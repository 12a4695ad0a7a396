#!/bin/bash
# HBM traffic of the hot-path kernels from the TCC counters (MI355X_MICROARCH.md "HBM"): separate --pmc passes
# for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), kernel-trace only.  Units are KiB; on gfx950
# FETCH_SIZE counts 128-B read requests at 64 B, so reads of wide coalesced streams are doubled by
# tools/traffic_summary.py before they are compared with byte counts.  The wrapped command is
# `bench.py --only-steps` (exactly WARMUP + STEPS steps, nothing else), so every kernel's launch count divided by the
# step count is its launches per step, and the sum over kernels is the traffic of one whole step (`__step__`).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
MODEL=${MODEL:-combsub}; STEPS=3; WARMUP=1
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$O/traffic_$c"
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d "$O/traffic_$c" -o t -- python "$R/bench.py" --model $MODEL --only-steps --steps $STEPS --warmup $WARMUP ${BENCH_ARGS:-} > "$O/traffic_$c.log" 2>&1
  tail -1 "$O/traffic_$c.log"
  f=$(find "$O/traffic_$c" -name "*.db" | head -1)
  [ -n "$f" ] && python "$R/tools/rocpd_pmc.py" "$f" 2>/dev/null | grep ddsp > "$O/traffic_$c.txt"
  rm -rf "$O/traffic_$c"
done
python "$R/tools/traffic_summary.py" --model $MODEL --steps $((STEPS + WARMUP)) "$O/traffic_FETCH_SIZE.txt" "$O/traffic_WRITE_SIZE.txt" | tee "$O/traffic.json"
