#!/bin/bash
# round 3, call R: backward of the short-time spectral filter with the next stage's inputs prefetched: head against the tree at
# 3 and 2 waves per SIMD (knob STFT_WPS; the forward kernel takes the same knob: 2 is its default)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
for k in combsubsuperfast; do
  for tag in head cur3 cur2 head cur3 cur2; do
    case $tag in
      head) env="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_head.so";;
      cur3) env="X=1";;
      cur2) env="DDSP_HIP_STFT_WPS=2";;
    esac
    echo -n "$tag: "; env $env timeout 120 python tools/train_step_probe.py $k 2>&1 | tail -1
  done
done
timeout 600 python -m pytest tests/test_backward_fast.py tests/test_parity_fast.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
