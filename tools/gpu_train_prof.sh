#!/bin/bash
# kernel stats of the training steps (forward + backward of the DSP tails; PROBE_ARGS=loss: with the reference's RSSLoss as the
# objective), one-stream order
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-v13}
cd /tmp
for k in ${MODELS:-combsub sins combsubsuperfast}; do
  DDSP_HIP_ONE_STREAM=1 timeout ${PROF_TIMEOUT:-75} rocprofv3 --kernel-trace -d "$O/tp_$k" -o t -- python "$R/tools/train_step_probe.py" $k ${PROBE_ARGS:-} > "$O/tp_$k.log" 2>&1
  tail -1 "$O/tp_$k.log"
  python "$R/tools/rocpd_stats.py" $(find "$O/tp_$k" -name "*.db" | head -1) 2>&1 | head -16 > "$O/${V}_train_${k}_kernel_stats.csv"
  rm -rf "$O/tp_$k"
  head -8 "$O/${V}_train_${k}_kernel_stats.csv"
done
