#!/bin/bash
# kernel timeline of one steady-state step (two streams, then one stream): start offset, duration, gap before each kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-gaps}; M=${MODEL:-combsub}
cd /tmp
for mode in two one; do
  rm -rf "$O/gp"
  if [ $mode = one ]; then export DDSP_HIP_ONE_STREAM=1; else unset DDSP_HIP_ONE_STREAM; fi
  timeout 300 rocprofv3 --kernel-trace -d "$O/gp" -o g -- python "$R/bench.py" --model $M --only-steps --steps 12 --warmup 3 > "$O/gp_$mode.log" 2>&1
  f=$(find "$O/gp" -name "*.db" | head -1)
  echo "== $M, $mode stream(s)" | tee "$O/${V}_${M}_gaps_$mode.txt"
  python "$R/tools/rocpd_gaps.py" "$f" 2>&1 | tee -a "$O/${V}_${M}_gaps_$mode.txt"
  rm -rf "$O/gp"
done
