#!/bin/bash
# round 3, call K: the loss straight from the waveforms (csrc/loss_czt.hip): parity tests, the rssloss step with the
# in-kernel chirp-z STFT and with torch.stft (same box), per-kernel times of both
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03k}
timeout 600 python -m pytest tests/test_loss.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee "$O/${V}_pytest_loss.log"
timeout 300 python bench.py --model rssloss --steps 50 2>&1 | tail -1 > "$O/${V}_bench_rssloss_czt.json"
DDSP_HIP_LOSS_TORCH_STFT=1 timeout 300 python bench.py --model rssloss --steps 50 2>&1 | tail -1 > "$O/${V}_bench_rssloss_torch.json"
cd /tmp
for tag in czt torch; do
  if [ $tag = torch ]; then export DDSP_HIP_LOSS_TORCH_STFT=1; fi
  rm -rf "$O/kp"; timeout 200 rocprofv3 --kernel-trace -d "$O/kp" -o k -- python "$R/bench.py" --model rssloss --steps 20 --warmup 3 > "$O/kp.log" 2>&1
  python "$R/tools/rocpd_stats.py" $(find "$O/kp" -name "*.db" | head -1) 2>&1 | head -24 > "$O/${V}_rssloss_${tag}_kernel_stats.csv"; rm -rf "$O/kp"
done
cd "$R"
python - <<'PY'
import json, glob, os
V = os.environ.get("V", "r03k")
for f in sorted(glob.glob("gpurun_out/%s_bench_*.json" % V)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), round(d["ms_per_step"], 4), d.get("eager_composition"))
    except Exception as e:
        print(f, "unreadable", e, open(f).read()[-600:])
PY
head -14 "$O/${V}_rssloss_czt_kernel_stats.csv"; head -10 "$O/${V}_rssloss_torch_kernel_stats.csv"
