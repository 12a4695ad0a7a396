#!/usr/bin/env python
"""Time individual hot-path kernels on the MI355X with events on the launch stream (B=32, 10 s, n=256)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddsp_svc_amd import _ffi, core, synth

SR, HOP = 44100, 512


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda:0")
    B, F, n = 32, 862, 256
    T, N = F * HOP, 2 * (n - 1)
    L = _ffi.lib()
    st = torch.cuda.current_stream().cuda_stream
    x = torch.rand(B, T, device=dev) * 2 - 1
    taps = torch.randn(B, F, N, device=dev) / N ** 0.5
    y = torch.empty(B, T, device=dev)
    res = {}
    flops = 4.0 * N * B * T
    for impl in [2, 3] + [int(v) for v in os.environ.get('FIR_EXTRA', '').split()]:
        ms = timeit(lambda: _ffi.check(L.ddsp_hip_fft_convolve(x.data_ptr(), 0, taps.data_ptr(), None, y.data_ptr(), None,
                                                               B, F, HOP, N, impl, st)))
        res["fir_impl%d_ms" % impl] = ms
        res["fir_impl%d_TFLOPs" % impl] = flops / ms / 1e9
    c = torch.randn(B, F, n, device=dev)
    tab = core.ir_table(n, dev)
    tp = torch.empty(B, F, N, device=dev)
    hw = torch.rand(B * F, device=dev) * 300 + 50
    re = torch.empty(B * F, n, device=dev)
    im = torch.empty(B * F, n, device=dev)
    res["allpass_ms"] = timeit(lambda: _ffi.check(L.ddsp_hip_allpass_response(c.data_ptr(), n, B * F, n, re.data_ptr(), im.data_ptr(), st)))
    res["ir_real_exp_hann_ms"] = timeit(lambda: _ffi.check(L.ddsp_hip_impulse_response(
        c.data_ptr(), n, None, 0, 1, 1.0 / 128, 1, None, B * F, n, tab.data_ptr(), tp.data_ptr(), st)))
    res["ir_real_exp_dyn_ms"] = timeit(lambda: _ffi.check(L.ddsp_hip_impulse_response(
        c.data_ptr(), n, None, 0, 1, 1.0, 2, hw.data_ptr(), B * F, n, tab.data_ptr(), tp.data_ptr(), st)))
    res["ir_complex_roll_ms"] = timeit(lambda: _ffi.check(L.ddsp_hip_impulse_response(
        re.data_ptr(), n, im.data_ptr(), n, 0, 1.0, 0, None, B * F, n, tab.data_ptr(), tp.data_ptr(), st)))
    f0 = torch.rand(B, F, device=dev) * 300 + 100
    ph = synth.phase(f0, SR, HOP)
    res["phase_ms"] = timeit(lambda: synth.phase(f0, SR, HOP))
    res["combtooth_ms"] = timeit(lambda: synth.combtooth(f0, ph, SR, HOP))
    ca = torch.randn(B, F, 256, device=dev)
    res["sins_bank_ms"] = timeit(lambda: synth.sinusoid_bank(f0, ph, ca, SR, HOP), reps=5, warm=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
