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
