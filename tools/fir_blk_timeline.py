#!/usr/bin/env python
"""Diagnostics: build libddsp_hip with -DDDSP_HIP_TIMELINE into a scratch dir, run k_fir_blk once at the headline shape
and print when its workgroups start, how long their prologue and their pairs take and when they end (wall_clock64
stamps, 100 MHz, one epoch for the whole chip)."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from ddsp_svc_amd import _ffi, build

so = os.environ.get("DDSP_TL_LIB") or os.path.join(ROOT, "tools", "ab", "libddsp_hip_tl.so")      # git-ignored; travels with gpurun's snapshot
if "--build" in sys.argv or not os.path.exists(so):
    out = "/tmp/ddsp_timeline"
    os.makedirs(out, exist_ok=True)
    os.makedirs(os.path.dirname(so), exist_ok=True)
    procs, objs = [], []
    for src in build.SOURCES:
        o = os.path.join(out, src.replace(".hip", ".o"))
        procs.append(subprocess.Popen([build._hipcc(), *build.FLAGS, "-DDDSP_HIP_TIMELINE", "-c", os.path.join(build.CSRC, src), "-o", o],
                                      stderr=subprocess.DEVNULL))
        objs.append(o)
    assert all(p.wait() == 0 for p in procs)
    subprocess.run([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", so], check=True)
    if "--build" in sys.argv:
        sys.exit(0)
L = _ffi.bind(ctypes.CDLL(so))
for kv in sys.argv[1:]:
    if "=" in kv:
        k, v = kv.split("=")
        assert L.ddsp_hip_set_tuning(k.encode(), int(v)) == 0, kv
L.ddsp_hip_debug_set_blk_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
B, F, n, HOP = 64, 431, 256, 512
T, N = F * HOP, 2 * (n - 1)
st = torch.cuda.current_stream().cuda_stream
x = torch.rand(B, T, device=dev) * 2 - 1
taps = torch.randn(B, F, N, device=dev) / N ** 0.5
y = torch.empty(B, T, device=dev)
for _ in range(3):
    L.ddsp_hip_fft_convolve(x.data_ptr(), 0, taps.data_ptr(), None, y.data_ptr(), None, B, F, HOP, N, 5, st)
nwg = 4096
tl = torch.zeros(nwg, 32, dtype=torch.int64, device=dev)
L.ddsp_hip_debug_set_blk_timeline(tl.data_ptr(), st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
L.ddsp_hip_fft_convolve(x.data_ptr(), 0, taps.data_ptr(), None, y.data_ptr(), None, B, F, HOP, N, 5, st)
e1.record()
torch.cuda.synchronize()
print("launch (events): %.1f us" % (e0.elapsed_time(e1) * 1e3))
traw = tl.cpu().numpy()
live = traw[:, 0] > 0
raw = traw[live]
t = raw.astype(np.float64)
where = raw[:, 31]
cyc = t[:, 24:30].copy()
t[:, 24:] = 0
print("workgroups:", t.shape[0])
t0 = t[:, 0].min()
us = lambda a: (a - t0) / 100.0
start = us(t[:, 0])
last = np.where(t > 0, t, 0).max(axis=1)
end = us(last)
pct = lambda a: "min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f" % (a.min(), np.percentile(a, 10), np.percentile(a, 50), np.percentile(a, 90), a.max())
print("start  [us]:", pct(start))
print("end    [us]:", pct(end))
print("life   [us]:", pct(end - start))
print("twiddle init [us]:", pct((t[:, 1] - t[:, 0]) / 100.0))
print("prologue     [us]:", pct((t[:, 2] - t[:, 1]) / 100.0))
xcc = (where >> 32) & 0xF
hw = where & 0xFFFFFFFF
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
life = end - start
print("life by XCC:", " ".join("%d:%.1f" % (x, life[(xcc == x) & (life > 60)].mean()) for x in np.unique(xcc)))
print("life by blockIdx %% 8:", " ".join("%.1f" % life[(np.arange(len(life)) % 8 == x) & (life > 60)].mean() for x in range(8)))
print("life by SE:", " ".join("%d:%.1f" % (x, life[(se == x) & (life > 60)].mean()) for x in np.unique(se)))
print("life by CU id:", " ".join("%d:%.1f" % (x, life[(cu == x) & (life > 60)].mean()) for x in np.unique(cu)))
print("life by first SIMD:", " ".join("%d:%.1f" % (x, life[(simd == x) & (life > 60)].mean()) for x in np.unique(simd)))
key = xcc * 4096 + se * 512 + sh * 256 + cu
u, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
print("CUs seen:", len(u), "workgroups per CU:", np.unique(cnt, return_counts=True))
cu_mean = np.array([life[(inv == i) & (life > 60)].mean() for i in range(len(u))])
print("per-CU mean life [us]:", pct(cu_mean[np.isfinite(cu_mean)]))
bid = np.nonzero(live)[0]
for i in range(2):
    print("blockIdx of the workgroups of one CU:", bid[inv == i], "pairs", (t[inv == i, 3:24] > 0).sum(axis=1))
srt = np.array([np.sort(life[inv == i]) for i in range(len(u)) if (inv == i).sum() == 4 and life[inv == i].min() > 60])
if len(srt):
    print("lifetimes of the 4 workgroups of a CU, sorted, mean over CUs [us]:", " ".join("%.1f" % v for v in srt.mean(axis=0)))
within = np.array([life[(inv == i) & (life > 60)].std() for i in range(len(u)) if ((inv == i) & (life > 60)).sum() > 1])
print("std of life within a CU: mean %.2f; std of per-CU means %.2f" % (within.mean(), np.nanstd(cu_mean)))
names = ["block pair (lockstep) + park", "spectral product", "inverse | next taps (lockstep) + park", "tap split", "ring + stores"]
ok = cyc[:, 5] > 0
for i, nm in enumerate(names):
    d = cyc[ok, i + 1] - cyc[ok, i]
    print("  %-30s %s cycles" % (nm, pct(d)))
npairs = (t[:, 3:24] > 0).sum(axis=1)
print("pairs per workgroup:", np.unique(npairs, return_counts=True))
it = np.diff(t[:, 2:24], axis=1)
it = it[(t[:, 3:24] > 0)] / 100.0
print("one pair     [us]:", pct(it))
for k in range(0, 16):
    col = t[:, 3 + k]; ok = col > 0
    if ok.any():
        d = (col[ok] - t[ok, 2 + k]) / 100.0
        print("  pair %2d: mean %.2f us (n=%d)" % (k, d.mean(), ok.sum()))
L.ddsp_hip_debug_set_blk_timeline(None, st)
