#!/usr/bin/env python
"""Diagnostics (timeline build): which workgroups of the dynamic-window tap synthesis have a slow stage C, and does it follow
their rows' half widths, their index, or their round?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ddsp_svc_amd import _ffi, core

so = os.environ.get("TL_LIB", os.path.join(ROOT, "tools", "ab", "libddsp_hip_tl.so"))
L = _ffi.bind(ctypes.CDLL(so))
L.ddsp_hip_debug_set_pfa_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
B, F, n = 64, 431, 256
rows, N = B * F, 2 * (n - 1)
st = torch.cuda.current_stream().cuda_stream
tab = core.ir_table(n, dev)
c = torch.randn(rows, n, device=dev) * 0.7
taps = torch.empty(rows, N, device=dev)
p = lambda t: t.data_ptr()
nwg = (rows + 15) // 16
pct = lambda a: "min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max())
for name, hw in (("random 138..827", (1.5 * 44100.0) / (torch.rand(rows, device=dev) * 400 + 80)),
                 ("constant 300", torch.full((rows,), 300.0, device=dev)),
                 ("constant 100", torch.full((rows,), 100.0, device=dev)),
                 ("constant 1000 (never clamps)", torch.full((rows,), 1000.0, device=dev))):
    fn = lambda: L.ddsp_hip_impulse_response(p(c), n, None, 0, 1, 1.0, 2, p(hw), rows, n, p(tab), p(taps), st)
    for _ in range(3):
        assert fn() == 0
    tl = torch.zeros(nwg, 8, dtype=torch.int64, device=dev)
    L.ddsp_hip_debug_set_pfa_timeline(tl.data_ptr(), st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    L.ddsp_hip_debug_set_pfa_timeline(None, st)
    t = tl.cpu().numpy().astype(np.float64)
    t0 = t[:, 0].min()
    sc = (t[:, 4] - t[:, 3]) / 100
    start = (t[:, 0] - t0) / 100
    print("== %s: launch %.1f us; stage C %s" % (name, e0.elapsed_time(e1) * 1e3, pct(sc)))
    slow = sc > 6.0
    print("   slow workgroups: %d of %d; of the first round (start < 2 us): %d of %d; their index: %s ..." % (
        slow.sum(), nwg, (slow & (start < 2)).sum(), (start < 2).sum(), np.nonzero(slow)[0][:24]))
    if slow.any():
        h = hw.cpu().numpy()
        hmin = np.array([h[16 * i:16 * i + 16].min() for i in range(nwg)])
        print("   min half width of slow workgroups: %s | of all: %s" % (pct(hmin[slow]), pct(hmin)))
        print("   stage C start (us after launch) of slow ones: %s | of all: %s" % (pct(((t[:, 3] - t0) / 100)[slow]), pct((t[:, 3] - t0) / 100)))
