#!/bin/bash
# round 3, call M: same-box A/B of the loss kernels: single-frame forward (commit 'Spectral loss straight from the waveforms',
# tools/ab/libddsp_hip_czt1.so) against the two-frame lockstep forward; one round of resident workgroups
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03m}
for tag in czt1 cur czt1 cur; do
  if [ $tag = cur ]; then lib="DDSP_HIP_CZT_ROUNDS=1"; else lib="DDSP_HIP_LIB=$R/tools/ab/libddsp_hip_$tag.so"; fi
  env $lib timeout 200 python bench.py --model rssloss --steps 60 2>&1 | tail -1 > "$O/${V}_rss_ab.json"
  python - "$tag" <<'PY'
import json, sys, os
d = json.loads(open("gpurun_out/%s_rss_ab.json" % os.environ.get("V", "r03m")).read().strip().splitlines()[-1])
print(sys.argv[1], "step ms %.4f" % d["ms_per_step"], "fwd+bwd alone %.4f" % d["roofline"]["avg_ms"], "forward only %.4f" % d["roofline"]["forward_only_ms"])
PY
done
