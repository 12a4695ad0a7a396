#!/bin/bash
# code size (bytes) of every kernel of a translation unit: hipcc device pass -> llvm-readelf symbol sizes (no GPU needed)
#     tools/code_sizes.sh ddsp_svc_amd/csrc/loss_czt.hip [extra flags]
set -eu
R=/root/repo; SRC=$1; shift
W=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --cuda-device-only -I$R/include -I$R/ddsp_svc_amd/csrc "$@" -c $SRC -o $W/k.o 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$W/k.o --output=$W/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf -sW $W/k.co | awk '$4=="FUNC" && $7!="UND"{print $3, $8}' | sort -u | sort -n | while read sz name; do echo "$sz $(echo $name | c++filt | cut -c1-100)"; done
rm -rf $W
