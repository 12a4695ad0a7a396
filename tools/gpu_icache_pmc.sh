#!/bin/bash
# instruction-cache counters of the steps' kernels (is a long unrolled loop body running out of the instruction cache?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-icache}
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\bSQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_CACHE)[A-Z0-9_]*" | sort -u | tee "$O/${V}_icache_counters.txt"
: > "$O/${V}_icache_pmc.txt"
for m in ${MODELS:-combsub combsubfast combsubsuperfast}; do
  rm -rf "$O/ipmc"
  DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --pmc ${COUNTERS:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES} --kernel-trace -d "$O/ipmc" -o p -- python "$R/bench.py" --model $m --only-steps --steps 3 --warmup 1 > "$O/ipmc_$m.log" 2>&1
  f=$(find "$O/ipmc" -name "*.db" | head -1)
  echo "== $m" >> "$O/${V}_icache_pmc.txt"
  [ -n "$f" ] && python "$R/tools/rocpd_pmc.py" "$f" 2>/dev/null | grep "ddsp::" >> "$O/${V}_icache_pmc.txt"
  tail -3 "$O/ipmc_$m.log" | head -2
  rm -rf "$O/ipmc"
done
cat "$O/${V}_icache_pmc.txt"
