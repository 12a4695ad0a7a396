#!/bin/bash
# round 3: why part of the pool is slow -- instruction-fetch counters of the CombSub step's kernels (one-stream order), round-2
# library (three tap-synthesis code objects, 25 KB filter, 15 KB phase kernel: ~93 KB per step) against the current one (~45 KB),
# with the prev / cur step times that tell which kind of box this is
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03_icache}
B="python bench.py --no-cpu-baseline --no-module-mode --no-live-traffic --no-also --steps 60"
OUT="$O/${V}_box_compare.txt"
: > "$OUT"
for tag in prev cur; do
  if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
  ms=$(env $lib timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])")
  ms1=$(env $lib DDSP_HIP_ONE_STREAM=1 timeout 300 $B 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f' % d['ms_per_step'])")
  echo "== $tag: CombSub step $ms ms (two streams), $ms1 ms (one stream)" | tee -a "$OUT"
done
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk|socclk" | head -8 >> "$OUT"
cd /tmp
for tag in prev cur; do
  if [ $tag = cur ]; then lib=X=1; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
  echo "== $tag: per-kernel counters, average per launch (4 steps, one stream)" >> "$OUT"
  for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
             "SQC_TC_INST_REQ SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"; do
    rm -rf "$O/ipmc"
    env $lib DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --pmc $set --kernel-trace -d "$O/ipmc" -o p -- python "$R/bench.py" --only-steps --steps 3 --warmup 1 > "$O/ipmc.log" 2>&1
    f=$(find "$O/ipmc" -name "*.db" | head -1)
    [ -n "$f" ] && python "$R/tools/rocpd_pmc.py" "$f" 2>/dev/null | grep "ddsp::" >> "$OUT"
    rm -rf "$O/ipmc"
  done
done
cat "$OUT" | grep -E "^==|k_fir_blk|k_taps" | grep -E "^==|IFETCH|ICACHE_MISSES |TC_INST|GRBM" 
