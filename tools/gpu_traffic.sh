#!/bin/bash
# HBM traffic of the hot-path kernels from the TCC counters (MI355X_MICROARCH.md "HBM"): separate --pmc passes
# for FETCH_SIZE and WRITE_SIZE (they do not fit one pass), kernel-trace only.  Units are KiB; on gfx950
# FETCH_SIZE counts 128-B read requests at 64 B, so reads of wide coalesced streams are doubled by
# tools/traffic_summary.py before they are compared with byte counts.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$O/traffic_$c"
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d "$O/traffic_$c" -o t -- python "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$O/traffic_$c.log" 2>&1
  tail -1 "$O/traffic_$c.log"
  f=$(find "$O/traffic_$c" -name "*.db" | head -1)
  [ -n "$f" ] && python "$R/tools/rocpd_pmc.py" "$f" 2>/dev/null | grep ddsp > "$O/traffic_$c.txt"
  rm -rf "$O/traffic_$c"
done
python "$R/tools/traffic_summary.py" "$O/traffic_FETCH_SIZE.txt" "$O/traffic_WRITE_SIZE.txt" | tee "$O/traffic.json"
