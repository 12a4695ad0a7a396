#!/bin/bash
# build a variant library tools/ab/libddsp_hip_<tag>.so from the working tree with extra flags for ONE translation unit
# (SRC=ir_pfa tools/build_variant.sh rows14 -DDDSP_PFA_ROWS=14; default SRC=fir_blk); the other objects are the in-tree build's.
# -DDDSP_AB_GENERATIONS compiles the superseded kernel generations back in (k_fir_blk<2>, k_fir_fft<false>, k_sins_bank2,
# k_sins_bank2_bwd) and makes their knobs (BLK_WPS, BLK_PADLDS, SINS_V1 = 2) live: the product library ships one generation
# per kernel, so a same-box A/B against an older one is  SRC=fir_blk tools/build_variant.sh gen -DDDSP_AB_GENERATIONS
set -eu
TAG=$1; shift
SRC=${SRC:-fir_blk}
R=/root/repo; W=$(mktemp -d)
cd $R/ddsp_svc_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$R/include "$@" -c $SRC.hip -o $W/$SRC.o 2>/dev/null
objs=""
for f in $R/ddsp_svc_amd/lib/*.o; do b=$(basename $f); [ $b = $SRC.o ] || objs="$objs $f"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $W/$SRC.o -o $R/tools/ab/libddsp_hip_$TAG.so
rm -rf $W; ls -la $R/tools/ab/libddsp_hip_$TAG.so
