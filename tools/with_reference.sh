#!/bin/bash
# Run a gpurun command with the reference checkout present on the GPU box -- ONCE per round, for the tests that hold the
# drop-in modules against the reference's own classes on the MI355X (tests/test_modules.py, test_backward_*.py, test_mel.py)
# and for bench.py's `cpu_baseline.kind = "reference"` / cfg-5 lines.
#
# /root/reference does not exist on the GPU box and reference SOURCES are never committed: this script puts a throw-away copy
# under build/reference_checkout (build/ is git-ignored, so it travels with the gpurun snapshot but never enters history),
# runs the command with DDSP_REFERENCE_PATH pointing at it, and removes the copy again whatever happens.
#     tools/with_reference.sh 1500 'V=r04_ref bash tools/gpu_r04.sh reference'
set -u
R=/root/repo
LIMIT=${1:?seconds}; shift
CMD=${1:?command}
[ -d /root/reference/ddsp ] || { echo "no /root/reference here"; exit 2; }
mkdir -p "$R/build"
rm -rf "$R/build/reference_checkout"
cp -r /root/reference "$R/build/reference_checkout"
find "$R/build/reference_checkout" -name "*.png" -delete
trap 'rm -rf "$R/build/reference_checkout"' EXIT
git -C "$R" check-ignore -q build/reference_checkout || { echo "build/ is not git-ignored: refusing"; exit 3; }
/usr/local/graft/bin/gpurun --timeout "$LIMIT" -- "export DDSP_REFERENCE_PATH=\$GRAFT_REPO_ROOT/build/reference_checkout; $CMD"
