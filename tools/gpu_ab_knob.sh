#!/bin/bash
# same-box A/B of a launcher knob: the CombSub step (one-stream and default) for each value, + kernel traces
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-ab5}; KNOB=${KNOB:-DDSP_HIP_PFA_TR}; VALUES=${VALUES:-"8 2"}
timeout 900 python -m pytest tests/test_parity.py tests/test_baseline_shapes.py -m gpu -x -q -k "prime_factor or impulse or tail or cfg1 or filters" 2>&1 | tail -3 | tee "$O/${V}_pytest_subset.log"
B="python bench.py --model ${MODEL:-combsub} --no-cpu-baseline --no-module-mode --no-live-traffic --no-also"
for rep in 1 2; do
  for v in $VALUES; do
    env $KNOB=$v DDSP_HIP_ONE_STREAM=1 timeout 300 $B 2>&1 | tail -1 > "$O/${V}_bench_${v}_one_$rep.json"
    env $KNOB=$v timeout 300 $B 2>&1 | tail -1 > "$O/${V}_bench_${v}_two_$rep.json"
  done
done
cd /tmp
for v in $VALUES; do
  env $KNOB=$v DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$O/prof_ab5" -o ab -- python "$R/bench.py" --model ${MODEL:-combsub} --only-steps --steps 20 --warmup 3 > "$O/prof_ab5.log" 2>&1
  python "$R/tools/rocpd_stats.py" $(find "$O/prof_ab5" -name "*.db" | head -1) 2>&1 | head -8 > "$O/${V}_${v}_kernel_stats.csv"
  rm -rf "$O/prof_ab5"
done
cd "$R"
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "ab5")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), round(d.get("ms_per_step_events") or 0, 4), "%.3e" % d["value"])
    except Exception as e:
        print(f, "ERR", e, open(f).read()[-300:])
PY
for v in $VALUES; do cat "$O/${V}_${v}_kernel_stats.csv"; done
