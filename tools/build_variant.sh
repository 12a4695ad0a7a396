#!/bin/bash
# build a variant library tools/ab/libddsp_hip_<tag>.so from the working tree with extra flags for fir_blk.hip only
set -eu
TAG=$1; shift
R=/root/repo; W=$(mktemp -d)
cd $R/ddsp_svc_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$R/include "$@" -c fir_blk.hip -o $W/fir_blk.o 2>/dev/null
objs=""
for f in $R/ddsp_svc_amd/lib/*.o; do b=$(basename $f); [ $b = fir_blk.o ] || objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $W/fir_blk.o -o $R/tools/ab/libddsp_hip_$TAG.so
rm -rf $W; ls -la $R/tools/ab/libddsp_hip_$TAG.so
