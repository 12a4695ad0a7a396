#!/bin/bash
# PMC counters of the loss kernels (tools/loss_steps.py), two passes of 8 counters each, kernel trace only beside them
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
cd /tmp
: > "$O/loss_pmc.txt"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rm -rf "$O/lpmc"
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d "$O/lpmc" -o p -- python "$R/tools/loss_steps.py" 3 > "$O/lpmc_$i.log" 2>&1
  f=$(find "$O/lpmc" -name "*.db" | head -1)
  [ -n "$f" ] && python "$R/tools/rocpd_pmc.py" "$f" 2>/dev/null | grep "ddsp::" >> "$O/loss_pmc.txt"
  rm -rf "$O/lpmc"
done
cat "$O/loss_pmc.txt"
