#!/usr/bin/env python
"""Tap synthesis and its adjoint at bin counts other than 256, B = 32 x 862 frames: the chirp-z kernels (csrc/ir_czt.hip) against
the dense MFMA contraction they replace (knob TAPS_GEMM = 1), beside the prime-factor kernel at 256 bins."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddsp_svc_amd import _ffi, core

dev = torch.device("cuda:0")
B, F = 32, 862
g = torch.Generator().manual_seed(3)
lib = _ffi.lib()


def ms_of(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {}
for n in (128, 256, 257, 512, 1025):
    N = 2 * (n - 1)
    mag = torch.exp(torch.randn(B, F, n, generator=g)).to(dev)
    z = torch.complex(torch.randn(B, F, n, generator=g), torch.randn(B, F, n, generator=g)).to(dev)
    hw = (torch.rand(B, F, 1, generator=g) * 300 + 20).to(dev)
    go = torch.randn(B, F, N, generator=g).to(dev)
    cases = {"hann (real)": lambda r: core.frequency_impulse_response(r[0]),
             "dynamic (real)": lambda r: core.frequency_impulse_response(r[0], half_width_frames=hw),
             "roll (complex)": lambda r: core.frequency_impulse_response(r[1], hann_window=False)}
    for name, fn in cases.items():
        row = {}
        for form in ("fast", "gemm"):
            if form == "gemm" and n == 256:
                continue
            lib.ddsp_hip_set_tuning(b"TAPS_GEMM", 1 if form == "gemm" else 0)
            row[form + "_ms"] = round(ms_of(lambda: fn((mag, z))), 4)
            a = mag.clone().requires_grad_(True)
            zz = z.clone().requires_grad_(True)
            t = fn((a, zz))
            row[form + "_adjoint_ms"] = round(ms_of(lambda: torch.autograd.grad(t, zz if "complex" in name else a, go, retain_graph=True)), 4)
        lib.ddsp_hip_set_tuning(b"TAPS_GEMM", 0)
        row["kernel"] = "k_taps_pfa510" if n == 256 else "k_taps_czt"
        cplx = 2 if "complex" in name else 1
        row["hbm_frac_fast"] = round(4.0 * (cplx * n + N) * B * F / (row["fast_ms"] * 1e-3) / 8e12, 4)
        out["n_mag %d, %s" % (n, name)] = row
        print("n_mag %d, %s" % (n, name), row, flush=True)
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r03_taps_forms.json"), "w"), indent=1)
