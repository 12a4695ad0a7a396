#!/bin/bash
# same-box A/B of two builds (tools/ab/libddsp_hip_prev.so against the in-tree library) on the Sins rows: step, kernel trace of the
# step, training step and its trace; the GPU suite with the in-tree library first
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
export V=${V:-r04_sins}
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3 | tee "$O/${V}_pytest_gpu.log"
lib_of() { if [ "$1" = cur ]; then echo "$R/ddsp_svc_amd/lib/libddsp_hip.so"; else echo "$R/tools/ab/libddsp_hip_$1.so"; fi; }
for rep in 1 2; do for t in prev cur; do
  for m in sins combsub; do
    DDSP_HIP_LIB=$(lib_of $t) timeout 300 python bench.py --model $m --no-cpu-baseline --no-module-mode --no-live-traffic --no-also 2>/dev/null | tail -1 > "$O/${V}_bench_${m}_${t}_$rep.json"
    python - "$O/${V}_bench_${m}_${t}_$rep.json" "$m $t" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read())
print(sys.argv[2], "ms", round(d["ms_per_step"], 4))
PY
  done
  for k in combsub sins; do echo "$t $(DDSP_HIP_LIB=$(lib_of $t) timeout 120 python tools/train_step_probe.py $k 2>&1 | tail -1)"; done
done; done | tee "$O/${V}_ab.txt"
for t in prev cur; do
  ( cd /tmp; rm -rf "$O/ps_$t"
    DDSP_HIP_LIB=$(lib_of $t) DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats -d "$O/ps_$t" -o t -- python "$R/bench.py" --model sins --only-steps --steps 20 --warmup 3 > /dev/null 2>&1
    python "$R/tools/rocpd_stats.py" $(find "$O/ps_$t" -name "*.db" | head -1) 2>&1 | head -8 > "$O/${V}_sins_${t}_kernel_stats.csv"; rm -rf "$O/ps_$t"
    DDSP_HIP_LIB=$(lib_of $t) DDSP_HIP_ONE_STREAM=1 timeout 120 rocprofv3 --kernel-trace -d "$O/pt_$t" -o t -- python "$R/tools/train_step_probe.py" sins > /dev/null 2>&1
    python "$R/tools/rocpd_stats.py" $(find "$O/pt_$t" -name "*.db" | head -1) 2>&1 | head -12 > "$O/${V}_train_sins_${t}_kernel_stats.csv"; rm -rf "$O/pt_$t" )
  echo "== $t"; cut -c1-100 "$O/${V}_sins_${t}_kernel_stats.csv"; cut -c1-100 "$O/${V}_train_sins_${t}_kernel_stats.csv"
done
