#!/bin/bash
# tools/ab/libddsp_hip_prev.so: the library of a commit (default HEAD), for the same-box A/B of gpu_ab_builds.sh
set -eu
R=$(cd "$(dirname "$0")/.." && pwd); REV=${1:-HEAD}; W=$(mktemp -d)
git -C "$R" archive "$REV" ddsp_svc_amd/csrc include | tar -x -C "$W"
cd "$W/ddsp_svc_amd/csrc"
for f in *.hip; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -c "$f" -o "${f%.hip}.o" 2>/dev/null & done
wait
mkdir -p "$R/tools/ab"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o "$R/tools/ab/libddsp_hip_prev.so"
rm -rf "$W"; ls -la "$R/tools/ab/libddsp_hip_prev.so"
