#!/bin/bash
# PMC counters of every kernel of the CombSub and Sins steps (one-stream order), two passes of 8 counters each
# (another row: MODELS=mel BENCH_ARGS="--keyshift 3 --no-cpu-baseline --no-also --steps 3 --warmup 1 --prewarm-seconds 0" TAG=mel_keyshift3)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
for m in ${MODELS:-combsub sins}; do
  : > "$O/step_pmc_$m.txt"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT"; do
    i=$((i+1))
    rm -rf "$O/spmc"
    DDSP_HIP_ONE_STREAM=1 timeout 300 rocprofv3 --pmc $set --kernel-trace -d "$O/spmc" -o p -- python "$R/bench.py" --model $m ${BENCH_ARGS:---only-steps --steps 3 --warmup 1} > "$O/spmc_$m$i.log" 2>&1
    f=$(find "$O/spmc" -name "*.db" | head -1)
    [ -n "$f" ] && python "$R/tools/rocpd_pmc.py" "$f" 2>/dev/null | grep "ddsp::" >> "$O/step_pmc_$m.txt"
    rm -rf "$O/spmc"
  done
  wc -l "$O/step_pmc_$m.txt"
done
head -60 "$O/step_pmc_combsub.txt"
