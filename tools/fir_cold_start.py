#!/usr/bin/env python
"""What makes k_fir_blk's first microseconds slow on some boxes?  The timeline build's stamps for the LAST of a sequence of
launches, for several sequences: the filter right after itself, after a tiny kernel, after another big kernel, after an
event hand-over from another stream.  Prints prologue and first-pair times (steady state: ~8 and ~6.3 us)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ddsp_svc_amd import _ffi, core

so = os.path.join(ROOT, "tools", "ab", "libddsp_hip_tl.so")
L = _ffi.bind(ctypes.CDLL(so))
L.ddsp_hip_debug_set_blk_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0")
B, F, n, HOP = 64, 431, 256, 512
T, N = F * HOP, 2 * (n - 1)
rows = B * F
main = torch.cuda.current_stream()
st = main.cuda_stream
side = torch.cuda.Stream()
x = torch.rand(B, T, device=dev) * 2 - 1
taps = torch.randn(B, F, N, device=dev) / N ** 0.5
y = torch.empty(B, T, device=dev)
c = torch.randn(rows, n, device=dev) * 0.7
tab = core.ir_table(n, dev)
t2 = torch.empty(rows, N, device=dev)
tl = torch.zeros(4096, 32, dtype=torch.int64, device=dev)
p = lambda t: t.data_ptr()
fir = lambda: L.ddsp_hip_fft_convolve(p(x), 0, p(taps), None, p(y), None, B, F, HOP, N, 5, st)
pfa = lambda: L.ddsp_hip_impulse_response(p(c), n, None, 0, 1, 1.0 / 128, 1, None, rows, n, p(tab), p(t2), st)
tiny = lambda: tl[4000:4001].add_(0)
big_torch = lambda: torch.sin_(y)


def cross():
    with torch.cuda.stream(side):
        side.wait_stream(main)
        tl[4001:4002].add_(0)
    main.wait_stream(side)


def report(name, seq):
    for _ in range(3):
        fir()
    torch.cuda.synchronize()
    L.ddsp_hip_debug_set_blk_timeline(p(tl), st)
    fir()
    for f in seq:
        f()
    tl[:4000].zero_()
    fir()
    torch.cuda.synchronize()
    L.ddsp_hip_debug_set_blk_timeline(None, st)
    t = tl[:4000].cpu().numpy().astype(np.float64)
    t = t[t[:, 0] > 0]
    pro = (t[:, 2] - t[:, 0]) / 100.0
    p0 = (t[:, 3] - t[:, 2]) / 100.0
    p1 = (t[:, 4] - t[:, 3]) / 100.0
    end = (np.where(t[:, :24] > 0, t[:, :24], 0).max(axis=1) - t[:, 0].min()) / 100.0
    print("%-46s prologue %5.1f  pair 0 %5.1f  pair 1 %5.1f  last workgroup done %6.1f us" % (name, np.median(pro), p0.mean(), p1.mean(), end.max()))


import time
seqs = [("filter, [zero stamps], filter", []), ("filter, tiny torch kernel, filter", [tiny]),
        ("filter, tap synthesis kernel, filter", [pfa]), ("filter, torch sin_ over 56 MB, filter", [big_torch]),
        ("filter, event hand-over via another stream, filter", [cross]), ("filter, tap synthesis + hand-over, filter", [pfa, cross])]
t_begin = time.time()
for rnd in range(3):
    for name, seq in (seqs if rnd != 1 else seqs[::-1]):
        print("t=%6.3f s " % (time.time() - t_begin), end="")
        report(name, seq)
