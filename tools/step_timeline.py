#!/usr/bin/env python
"""Run under `rocprofv3 --kernel-trace`: phase A = the fused ddsp_hip_combsub_synth call in a loop (one stream),
phase B = the same kernels through the per-operation C calls in the same order.  tools/rocpd_phases.py then prints
the per-kernel average of either phase, to see whether a kernel's duration depends on how it is launched."""
import os
import sys

os.environ["DDSP_HIP_ONE_STREAM"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ddsp_svc_amd import _ffi, core, synth

dev = torch.device("cuda:0")
SR, HOP, n = 44100, 512, 256
N = 2 * (n - 1)
B, F = 32, 862
T = F * HOP
L = _ffi.lib()
tab = core.ir_table(n, dev)
f0, ctrls, noise = bench.make_inputs("combsub", B, F, (n, n, n), dev, 1234)
c_gd, c_h, c_nz = ctrls
ld = c_gd.stride(1)
rows = B * F
hw = ((1.5 * SR) / (f0.reshape(-1) + 1e-3)).contiguous()
re = torch.empty(rows, n, device=dev)
im = torch.empty(rows, n, device=dev)
t1 = torch.empty(rows, N, device=dev)
comb, h1, h2, signal = (torch.empty(B, T, device=dev) for _ in range(4))
p = lambda t: t.data_ptr()
ck = _ffi.check
s = torch.cuda.current_stream().cuda_stream


def fused():
    st = synth.phase(f0, SR, HOP)
    return synth.combsub_synth(f0, st, c_gd, c_h, c_nz, noise, SR, HOP, want_components=False)[0]


def serial():
    st = synth.phase(f0, SR, HOP)
    ck(L.ddsp_hip_allpass_response(p(c_gd), ld, rows, n, p(re), p(im), s))
    ck(L.ddsp_hip_impulse_response(p(re), n, p(im), n, 0, 1.0, 0, None, rows, n, p(tab), p(t1), s))
    ck(L.ddsp_hip_combtooth(p(f0), None, p(st.phase0), B, F, HOP, float(SR), 1, p(comb), s))
    ck(L.ddsp_hip_fft_convolve(p(comb), 0, p(t1), None, p(h1), None, B, F, HOP, N, 0, s))
    ck(L.ddsp_hip_impulse_response(p(c_h), ld, None, 0, 1, 1.0, 2, p(hw), rows, n, p(tab), p(t1), s))
    ck(L.ddsp_hip_fft_convolve(p(h1), 0, p(t1), None, p(h2), None, B, F, HOP, N, 0, s))
    ck(L.ddsp_hip_impulse_response(p(c_nz), ld, None, 0, 1, 1.0 / 128, 1, None, rows, n, p(tab), p(t1), s))
    ck(L.ddsp_hip_fft_convolve(p(noise), 0, p(t1), p(h2), p(signal), None, B, F, HOP, N, 0, s))


for fn in (fused, serial, fused, serial):
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    torch.zeros(1 << 20, device=dev).cos_()          # marker between phases: at::native cos kernel
    torch.cuda.synchronize()
