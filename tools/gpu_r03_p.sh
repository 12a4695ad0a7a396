#!/bin/bash
# round 3, call P: the loss with the per-frame balance of the two signals and with overlapping frames: tests, the rssloss step
# (A/B against the build before: tools/ab/libddsp_hip_czt2.so if present), SSSLoss with overlap 0.75 against the torch.stft form
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O="$R/gpurun_out"; mkdir -p "$O"; export TMPDIR=/tmp
V=${V:-r03p}
timeout 900 python -m pytest tests/test_loss.py tests/test_fuzz.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|assert" | tail -5 | tee "$O/${V}_pytest_loss.log"
for i in 1 2; do
  timeout 200 python bench.py --model rssloss --steps 60 2>&1 | tail -1 > "$O/${V}_rss.json"
  python - <<'PY'
import json, os
d = json.loads(open("gpurun_out/%s_rss.json" % os.environ.get("V", "r03p")).read().strip().splitlines()[-1])
print("rssloss step ms %.4f" % d["ms_per_step"], "fwd+bwd alone %.4f" % d["roofline"]["avg_ms"], "forward only %.4f" % d["roofline"]["forward_only_ms"], d["eager_composition"]["grad_rel_rms"])
PY
done
python - <<'PY'
import time, torch, os
from ddsp_svc_amd import loss as L
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
xt = (torch.randn(32, 441344, generator=g) * 0.1).to(dev)
xp = (xt * 0.9 + 0.02 * torch.randn(32, 441344, generator=g).to(dev)).requires_grad_(True)
for n, ov in ((1024, 0.75), (2047, 0.5), (397, 0.75)):
    res = {}
    for tag in ("kernel", "torch.stft"):
        if tag == "torch.stft":
            os.environ["DDSP_HIP_LOSS_TORCH_STFT"] = "1"
        else:
            os.environ.pop("DDSP_HIP_LOSS_TORCH_STFT", None)
        f = L.SSSLoss(n, 1.0, ov).to(dev)
        def step():
            return torch.autograd.grad(f(xt, xp), xp)[0]
        for _ in range(3):
            gr = step()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            gr = step()
        torch.cuda.synchronize()
        res[tag] = ((time.perf_counter() - t0) / 10 * 1e3, gr, float(f(xt, xp)))
    a, b = res["kernel"], res["torch.stft"]
    per = lambda x: x[:, :441344 // n * n].reshape(-1, n)
    err = (per(a[1]) - per(b[1])).pow(2).mean(1).sqrt() / per(b[1]).pow(2).mean(1).sqrt()
    print("SSSLoss(%d, overlap %.2f) fwd+bwd: in-kernel %.3f ms, torch.stft form %.3f ms; loss %.7f / %.7f; gradient per-frame rel err median %.2e, above 1e-3: %.2e"
          % (n, ov, a[0], b[0], a[2], b[2], float(err.median()), float((err > 1e-3).float().mean())))
PY
