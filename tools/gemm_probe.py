#!/usr/bin/env python
"""k_ir_gemm timing vs row count (one / two workgroups per CU, the real workload) and window mode."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddsp_svc_amd import _ffi, core

dev = torch.device("cuda:0")
n = 256
N = 2 * (n - 1)
L = _ffi.lib()
st = torch.cuda.current_stream().cuda_stream
tab = core.ir_table(n, dev)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps * 1e3, 1)


res = {}
for rows in (64 * 256, 64 * 512, 32 * 862, 64 * 768, 64 * 1024):
    c = torch.randn(rows, n, device=dev)
    im = torch.randn(rows, n, device=dev)
    tp = torch.empty(rows, N, device=dev)
    hw = torch.rand(rows, device=dev) * 300 + 50
    res["rows%d_real_exp_roll_us" % rows] = timeit(lambda: _ffi.check(L.ddsp_hip_impulse_response(
        c.data_ptr(), n, None, 0, 1, 1.0, 0, None, rows, n, tab.data_ptr(), tp.data_ptr(), st)))
    res["rows%d_real_exp_hann_us" % rows] = timeit(lambda: _ffi.check(L.ddsp_hip_impulse_response(
        c.data_ptr(), n, None, 0, 1, 1.0 / 128, 1, None, rows, n, tab.data_ptr(), tp.data_ptr(), st)))
    res["rows%d_real_exp_dyn_us" % rows] = timeit(lambda: _ffi.check(L.ddsp_hip_impulse_response(
        c.data_ptr(), n, None, 0, 1, 1.0, 2, hw.data_ptr(), rows, n, tab.data_ptr(), tp.data_ptr(), st)))
    res["rows%d_complex_roll_us" % rows] = timeit(lambda: _ffi.check(L.ddsp_hip_impulse_response(
        c.data_ptr(), n, im.data_ptr(), n, 0, 1.0, 0, None, rows, n, tab.data_ptr(), tp.data_ptr(), st)))
print(json.dumps(res, indent=1))

# the same kernels on the step's real operands: strided torch.split views of one [B,F,768] control tensor and the
# half widths of the synthetic f0 curves
import bench
from ddsp_svc_amd import synth
B, F = 32, 862
f0, ctrls, noise = bench.make_inputs("combsub", B, F, (n, n, n), dev, 1234)
rows = B * F
tp = torch.empty(rows, N, device=dev)
hw = (1.5 * 44100.0) / (f0.reshape(-1) + 1e-3)
res2 = {}
for name, c in (("split1_ld768", ctrls[1]), ("split2_ld768", ctrls[2]), ("contig", ctrls[1].contiguous())):
    ld = c.stride(1)
    res2["step_real_exp_dyn_%s_us" % name] = timeit(lambda: _ffi.check(L.ddsp_hip_impulse_response(
        c.data_ptr(), ld, None, 0, 1, 1.0, 2, hw.data_ptr(), rows, n, tab.data_ptr(), tp.data_ptr(), st)))
    res2["step_real_exp_hann_%s_us" % name] = timeit(lambda: _ffi.check(L.ddsp_hip_impulse_response(
        c.data_ptr(), ld, None, 0, 1, 1.0 / 128, 1, None, rows, n, tab.data_ptr(), tp.data_ptr(), st)))
hw2 = torch.rand(rows, device=dev) * 300 + 50
cc = ctrls[1].contiguous()
res2["step_real_exp_dyn_contig_randhw_us"] = timeit(lambda: _ffi.check(L.ddsp_hip_impulse_response(
    cc.data_ptr(), n, None, 0, 1, 1.0, 2, hw2.data_ptr(), rows, n, tab.data_ptr(), tp.data_ptr(), st)))
print(json.dumps(res2, indent=1))
