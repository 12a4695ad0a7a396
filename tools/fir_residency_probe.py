#!/usr/bin/env python
"""Why is k_fir_blk6 ~8 % slower inside a step than alone?  The filter timed alone (HIP events around back-to-back launches,
B = 32 x 10 s, N = 510) in five settings:
    same     the same input / tap / output buffers every launch (what bench.py's `roofline` leg does: 170 MB, memory-side-cache resident)
    rot K    K rotating sets of buffers (K x 226 MB: every launch reads what left the caches K launches ago -> from HBM)
    fresh    as rot, but each launch's inputs were WRITTEN by the launch before it (a copy kernel), as inside a step
    taps     a tap synthesis between filter launches (the step's alternation of code objects; time of the filters only, by events per launch)
    flush    a 512 MB fill between filter launches (cold L2 + memory-side cache, warm code)
Prints the mean / median / min per setting."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from ddsp_svc_amd import _ffi, core

dev = torch.device("cuda:0")
B, F, n, HOP = int(os.environ.get("B", 32)), 862, 256, 512
T, N = F * HOP, 2 * (n - 1)
rows = B * F
L = _ffi.lib()
st = torch.cuda.current_stream().cuda_stream
K = 6
xs = [torch.rand(B, T, device=dev) * 2 - 1 for _ in range(K)]
tps = [torch.randn(B, F, N, device=dev) / N ** 0.5 for _ in range(K)]
ys = [torch.empty(B, T, device=dev) for _ in range(K)]
ctrl = torch.randn(rows, n, device=dev) * 0.7
tab = core.ir_table(n, dev)
big = torch.empty(128 << 20, dtype=torch.float32, device=dev)          # 512 MB
p = lambda t: t.data_ptr()


def fir(i, j=None, k=None):
    j = i if j is None else j
    k = i if k is None else k
    _ffi.check(L.ddsp_hip_fft_convolve(p(xs[i]), 0, p(tps[j]), None, p(ys[k]), None, B, F, HOP, N, 5, st))


def taps(j):
    _ffi.check(L.ddsp_hip_impulse_response(p(ctrl), n, None, 0, 1, 1.0 / 128, 1, None, rows, n, p(tab), p(tps[j]), st))


def timed(name, body, between=None, reps=24):
    for i in range(4):
        if between:
            between(i)
        body(i)
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for i in range(reps):
        if between:
            between(i)
        ev[i][0].record()
        body(i)
        ev[i][1].record()
    torch.cuda.synchronize()
    d = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
    print("%-34s mean %6.1f  med %6.1f  min %6.1f  max %6.1f us" % (name, d.mean(), np.median(d), d.min(), d.max()), flush=True)


def back_to_back(name, body, reps=24):
    for i in range(4):
        body(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        body(i)
    e1.record()
    torch.cuda.synchronize()
    print("%-34s %6.1f us per launch (back to back)" % (name, e0.elapsed_time(e1) * 1e3 / reps), flush=True)


for rnd in range(2):
    back_to_back("same buffers", lambda i: fir(0))
    back_to_back("rot 2", lambda i: fir(i % 2))
    back_to_back("rot 6", lambda i: fir(i % K))
    timed("same buffers, events per launch", lambda i: fir(0))
    timed("rot 6, events per launch", lambda i: fir(i % K))
    timed("x fresh from a copy kernel", lambda i: fir(i % K), between=lambda i: xs[i % K].copy_(xs[(i + 1) % K]))
    timed("taps fresh from the tap synthesis", lambda i: fir(0, i % K, 0), between=lambda i: taps(i % K))
    timed("taps + x fresh", lambda i: fir(i % K, i % K, i % K), between=lambda i: (taps(i % K), xs[i % K].copy_(xs[(i + 1) % K])))
    timed("512 MB fill between launches", lambda i: fir(0), between=lambda i: big.fill_(1.0))
    timed("out rotating only", lambda i: fir(0, 0, i % K))
    timed("x rotating only", lambda i: fir(i % K, 0, 0))
    timed("taps rotating only", lambda i: fir(0, i % K, 0))
