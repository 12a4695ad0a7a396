#!/bin/bash
# round 3, call L: rounds of resident workgroups for the chirp-z loss kernels (knob CZT_ROUNDS)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03l}
for r in 1 2 4 8 16 2 4; do
  DDSP_HIP_CZT_ROUNDS=$r timeout 200 python bench.py --model rssloss --steps 60 2>&1 | tail -1 > "$O/${V}_rss_rounds.json"
  python - "$r" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/%s_rss_rounds.json" % __import__("os").environ.get("V", "r03l")).read().strip().splitlines()[-1])
print("rounds", sys.argv[1], "step ms %.4f" % d["ms_per_step"], "fwd+bwd alone %.4f" % d["roofline"]["avg_ms"], "forward only %.4f" % d["roofline"]["forward_only_ms"])
PY
done
