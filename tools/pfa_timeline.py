#!/usr/bin/env python
"""Diagnostics: the timeline build (tools/fir_blk_timeline.py --build makes it) running the three tap-synthesis kernels of
the CombSub step once each at the headline shape; prints when workgroups start and end and how long each stage takes."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ddsp_svc_amd import _ffi, core

so = os.path.join(ROOT, "tools", "ab", "libddsp_hip_tl.so")
L = _ffi.bind(ctypes.CDLL(so))
L.ddsp_hip_debug_set_pfa_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
B, F, n = 64, 431, 256
rows, N = B * F, 2 * (n - 1)
st = torch.cuda.current_stream().cuda_stream
tab = core.ir_table(n, dev)
c = torch.randn(rows, n, device=dev) * 0.7
f0 = (1.5 * 44100.0) / (torch.rand(rows, device=dev) * 400 + 80)      # half widths in taps (vocoder.py:851)
taps = torch.empty(rows, N, device=dev)
p = lambda t: t.data_ptr()
scr = torch.empty(L.ddsp_hip_allpass_taps_scratch_bytes(rows, n) // 4 + 16, device=dev)
calls = {
    "magnitude + Hann (noise)": lambda: L.ddsp_hip_impulse_response(p(c), n, None, 0, 1, 1.0 / 128, 1, None, rows, n, p(tab), p(taps), st),
    "magnitude + f0 window (harmonic)": lambda: L.ddsp_hip_impulse_response(p(c), n, None, 0, 1, 1.0, 2, p(f0), rows, n, p(tab), p(taps), st),
    "all-pass": lambda: L.ddsp_hip_allpass_taps(p(c), n, rows, n, p(tab), p(taps), p(scr), scr.numel() * 4, st),
}
nwg = (rows + 15) // 16
pct = lambda a: "min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max())
for name, fn in calls.items():
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
    print("== %s: launch %.1f us (events), %d workgroups" % (name, e0.elapsed_time(e1) * 1e3, nwg))
    print("  start [us]:", pct((t[:, 0] - t0) / 100))
    print("  end   [us]:", pct((t[:, 4] - t0) / 100))
    print("  life  [us]:", pct((t[:, 4] - t[:, 0]) / 100))
    for i, nm in enumerate(["stage 0: rows in, activation", "stage A: gather + DFT-17", "stage B: DFT-30 + scatter", "stage C: window + store"]):
        print("  %-32s %s" % (nm, pct((t[:, i + 1] - t[:, i]) / 100)))
    first = (t[:, 0] - t0) / 100 < 2.0
    print("  first-round workgroups: %d, life %s" % (first.sum(), pct(((t[:, 4] - t[:, 0]) / 100)[first])))
