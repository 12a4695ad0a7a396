#!/usr/bin/env python
"""Time k_stft_filter variants on the MI355X (events on the launch stream): occupancy variant x run length,
for the CombSubSuperFast (win 2048) and CombSubFast (win 1024) geometries at B=32 x 10 s."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddsp_svc_amd import synth

HOP = 512


def timeit(fn, reps=10, warm=2):
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
    B, F = int(os.environ.get("B", 32)), 862
    T = F * HOP
    res = {}
    for win, occs in ((2048, tuple(int(v) for v in os.environ.get("WPS4", "3").split())),
                      (1024, tuple(int(v) for v in os.environ.get("WPS2", "2 3").split()))):
        n = win // 2 + 1
        exc = torch.randn(B, T, device=dev)
        nz = torch.randn(B, T, device=dev)
        c = torch.randn(B, F, 4 * n, device=dev)
        hm, hp, nm, nph = torch.split(c, [n] * 4, dim=-1)
        w = torch.hann_window(win, device=dev)
        for occ in occs:
            _ffi.set_tuning("STFT_WPS", occ)
            for run in os.environ.get("RUNS", "0").split():
                if run != "0":
                    _ffi.set_tuning("STFT_RUN", int(run))
                else:
                    _ffi.set_tuning("STFT_RUN", 0)
                ms = timeit(lambda: synth.stft_filter(exc, nz, hm, hp, nm, nph if win == 2048 else None, w, HOP,
                                                      pad_reflect=True, normalize=True))
                res["win%d_wps%d_run%s_ms" % (win, occ, run)] = round(ms, 4)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
