#!/usr/bin/env python
"""Diagnostics: build libddsp_hip with -DDDSP_HIP_TIMELINE into a scratch dir, run the FIR kernel once and
print per-phase cycle statistics from the s_memtime stamps of each workgroup."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from ddsp_svc_amd import _ffi, build

out = "/tmp/ddsp_timeline"
os.makedirs(out, exist_ok=True)
objs = []
for src in build.SOURCES:
    o = os.path.join(out, src.replace(".hip", ".o"))
    subprocess.run([build._hipcc(), *build.FLAGS, "-DDDSP_HIP_TIMELINE", "-c", os.path.join(build.CSRC, src), "-o", o], check=True)
    objs.append(o)
so = os.path.join(out, "libddsp_hip_tl.so")
subprocess.run([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", so], check=True)
L = _ffi.bind(ctypes.CDLL(so))
L.ddsp_hip_debug_set_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
B, F, n, HOP = 32, 862, 256, 512
T, N = F * HOP, 2 * (n - 1)
st = torch.cuda.current_stream().cuda_stream
x = torch.rand(B, T, device=dev) * 2 - 1
taps = torch.randn(B, F, N, device=dev) / N ** 0.5
y = torch.empty(B, T, device=dev)
impl = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for _ in range(2):
    L.ddsp_hip_fft_convolve(x.data_ptr(), 0, taps.data_ptr(), None, y.data_ptr(), None, B, F, HOP, N, impl, st)
nblk = 768 if impl == 2 else 512
tl = torch.zeros(nblk, 32, 8, dtype=torch.int64, device=dev)
L.ddsp_hip_debug_set_timeline(tl.data_ptr(), st)
L.ddsp_hip_fft_convolve(x.data_ptr(), 0, taps.data_ptr(), None, y.data_ptr(), None, B, F, HOP, N, impl, st)
torch.cuda.synchronize()
t = tl.cpu().numpy().astype(np.float64)
valid = t[:, :, 0] > 0
names = ["fill(zero+taps)+bar", "scatter+bar", "prefetch issue", "contract", "epilogue", "bar(next)"]
print("tiles stamped:", int(valid.sum()))
for i, nm in enumerate(names):
    d = (t[:, :, i + 1] - t[:, :, i])[valid & (t[:, :, i + 1] > 0)]
    if d.size:
        print("%-22s mean %9.0f  p10 %9.0f  p50 %9.0f  p90 %9.0f  (n=%d)" % (nm, d.mean(), np.percentile(d, 10), np.percentile(d, 50), np.percentile(d, 90), d.size))
tot = (t[:, :, 5] - t[:, :, 0])[valid & (t[:, :, 5] > 0)]
print("tile total             mean %9.0f" % tot.mean())
# every XCD has its own counter epoch: spans are only meaningful per workgroup
span = []
for w in range(t.shape[0]):
    v = t[w][t[w] > 0]
    if v.size:
        span.append(v.max() - v.min())
span = np.array(span)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
L.ddsp_hip_debug_set_timeline(None, st)
e0.record()
for _ in range(10):
    L.ddsp_hip_fft_convolve(x.data_ptr(), 0, taps.data_ptr(), None, y.data_ptr(), None, B, F, HOP, N, impl, st)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("per-workgroup span: mean %.0f  max %.0f counter ticks; kernel %.4f ms -> counter rate >= %.3f GHz if it is the shader clock"
      % (span.mean(), span.max(), ms, span.max() / ms / 1e6))
