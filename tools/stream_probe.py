#!/usr/bin/env python
"""Does running the independent noise branch of CombSub on a second HIP stream pay?  The step built from the
per-operation C-ABI calls: (a) one stream, (b) noise taps + noise FIR on a second stream, (c) the harmonic
magnitude taps there as well; against the fused ddsp_hip_combsub_synth call.  Results are compared for equality."""
import json
import os
import sys

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
state = synth.phase(f0, SR, HOP)
hw = ((1.5 * SR) / (f0.reshape(-1) + 1e-3)).contiguous()
re = torch.empty(rows, n, device=dev)
im = torch.empty(rows, n, device=dev)
t1, t2, t3 = (torch.empty(rows, N, device=dev) for _ in range(3))
comb, h1, nz_out, signal = (torch.empty(B, T, device=dev) for _ in range(4))
p = lambda t: t.data_ptr()
main = torch.cuda.current_stream()
aux = torch.cuda.Stream()
ck = _ffi.check


def ops(sm, sa):
    """sm: stream handle of the harmonic chain; sa: of the noise branch (may be the same)"""
    ck(L.ddsp_hip_allpass_response(p(c_gd), ld, rows, n, p(re), p(im), sm))
    ck(L.ddsp_hip_impulse_response(p(re), n, p(im), n, 0, 1.0, 0, None, rows, n, p(tab), p(t1), sm))
    ck(L.ddsp_hip_combtooth(p(f0), None, p(state.phase0), B, F, HOP, float(SR), 1, p(comb), sm))
    ck(L.ddsp_hip_fft_convolve(p(comb), 0, p(t1), None, p(h1), None, B, F, HOP, N, 0, sm))


def serial():
    s = main.cuda_stream
    ck(L.ddsp_hip_impulse_response(p(c_nz), ld, None, 0, 1, 1.0 / 128, 1, None, rows, n, p(tab), p(t3), s))
    ck(L.ddsp_hip_fft_convolve(p(noise), 0, p(t3), None, p(nz_out), None, B, F, HOP, N, 0, s))
    ops(s, s)
    ck(L.ddsp_hip_impulse_response(p(c_h), ld, None, 0, 1, 1.0, 2, p(hw), rows, n, p(tab), p(t2), s))
    ck(L.ddsp_hip_fft_convolve(p(h1), 0, p(t2), p(nz_out), p(signal), None, B, F, HOP, N, 0, s))


def forked(harm_taps_on_aux):
    s, a = main.cuda_stream, aux.cuda_stream
    aux.wait_stream(main)
    ck(L.ddsp_hip_impulse_response(p(c_nz), ld, None, 0, 1, 1.0 / 128, 1, None, rows, n, p(tab), p(t3), a))
    ck(L.ddsp_hip_fft_convolve(p(noise), 0, p(t3), None, p(nz_out), None, B, F, HOP, N, 0, a))
    if harm_taps_on_aux:
        ck(L.ddsp_hip_impulse_response(p(c_h), ld, None, 0, 1, 1.0, 2, p(hw), rows, n, p(tab), p(t2), a))
    ops(s, a)
    if not harm_taps_on_aux:
        ck(L.ddsp_hip_impulse_response(p(c_h), ld, None, 0, 1, 1.0, 2, p(hw), rows, n, p(tab), p(t2), s))
    main.wait_stream(aux)
    ck(L.ddsp_hip_fft_convolve(p(h1), 0, p(t2), p(nz_out), p(signal), None, B, F, HOP, N, 0, s))


def fused():
    return synth.combsub_synth(f0, state, c_gd, c_h, c_nz, noise, SR, HOP, noise_is_u01=False, want_components=False)[0]


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 4)


def fused_order(bufs):
    """the kernel order of ddsp_hip_combsub_synth on one stream, with the given buffer assignment"""
    s = main.cuda_stream
    re_, im_, ta, tb, tc, cmb, hh, sig = bufs
    ck(L.ddsp_hip_allpass_response(p(c_gd), ld, rows, n, p(re_), p(im_), s))
    ck(L.ddsp_hip_impulse_response(p(re_), n, p(im_), n, 0, 1.0, 0, None, rows, n, p(tab), p(ta), s))
    ck(L.ddsp_hip_combtooth(p(f0), None, p(state.phase0), B, F, HOP, float(SR), 1, p(cmb), s))
    ck(L.ddsp_hip_fft_convolve(p(cmb), 0, p(ta), None, p(hh), None, B, F, HOP, N, 0, s))
    ck(L.ddsp_hip_impulse_response(p(c_h), ld, None, 0, 1, 1.0, 2, p(hw), rows, n, p(tab), p(tb), s))
    hm = cmb if bufs_alias_harm[0] else h2
    ck(L.ddsp_hip_fft_convolve(p(hh), 0, p(tb), None, p(hm), None, B, F, HOP, N, 0, s))
    ck(L.ddsp_hip_impulse_response(p(c_nz), ld, None, 0, 1, 1.0 / 128, 1, None, rows, n, p(tab), p(tc), s))
    ck(L.ddsp_hip_fft_convolve(p(noise), 0, p(tc), p(hm), p(sig), None, B, F, HOP, N, 0, s))


h2 = torch.empty(B, T, device=dev)
bufs_alias_harm = [False]
ws = torch.empty(2 * B * T + rows * N + rows + 64, device=dev)          # carve_synth's layout
w_buf0, w_buf1 = ws[:B * T], ws[B * T:2 * B * T]
w_taps = ws[2 * B * T:2 * B * T + rows * N]
res = {}
try:
    ref = fused().clone()
    res["fused_ms"] = timeit(fused)
except TypeError:
    ref = None
for rnd in range(2):
    res["serial_ms_%d" % rnd] = timeit(serial)
    s0 = signal.clone()
    res["forked_noise_ms_%d" % rnd] = timeit(lambda: forked(False))
    s1 = signal.clone()
    res["forked_noise_harmtaps_ms_%d" % rnd] = timeit(lambda: forked(True))
    s2 = signal.clone()
for rnd in range(2):
    bufs_alias_harm[0] = False
    res["fusedorder_separate_ms_%d" % rnd] = timeit(lambda: fused_order((re, im, t1, t2, t3, comb, h1, signal)))
    res["fusedorder_sharedtaps_ms_%d" % rnd] = timeit(lambda: fused_order((re, im, t1, t1, t1, comb, h1, signal)))
    bufs_alias_harm[0] = True
    res["fusedorder_sharedtaps_harmalias_ms_%d" % rnd] = timeit(lambda: fused_order((re, im, t1, t1, t1, comb, h1, signal)))
    res["fusedorder_carved_ms_%d" % rnd] = timeit(lambda: fused_order(
        (w_buf0[:rows * n], w_buf0[rows * n:2 * rows * n], w_taps, w_taps, w_taps, w_buf0, w_buf1, signal)))
    res["fused_ms_again_%d" % rnd] = timeit(fused)
res["forked_equal_serial"] = bool((s0 == s1).all() and (s0 == s2).all())
if ref is not None:
    res["serial_vs_fused_maxdiff"] = float((s0 - ref).abs().max())
print(json.dumps(res, indent=1))
