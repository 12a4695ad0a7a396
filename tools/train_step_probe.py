#!/usr/bin/env python
"""forward + backward of the DSP tails (B=32 x 10 s) for rocprofv3 --kernel-trace: which kernels a training step spends
its time in."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ddsp_svc_amd import synth

torch.autograd.set_multithreading_enabled(False)       # backward on the calling thread: under rocprofv3 the interpreter has been
                                                        # seen to hang at exit with autograd's device worker thread alive
dev = torch.device("cuda:0")
B, F, n = 32, 862, 256
kind = sys.argv[1] if len(sys.argv) > 1 else "combsub"
if kind == "combsubsuperfast":
    f0, ctrls, noise = bench.make_inputs(kind, B, F, (1025,) * 4, dev, 1234)
    w = torch.hann_window(2048, device=dev)
else:
    f0, ctrls, noise = bench.make_inputs(kind, B, F, (n, n, n), dev, 1234)
c = [x.clone().requires_grad_(True) for x in ctrls]
R = torch.randn(B, F * 512, device=dev)


def step():
    if kind == "combsubsuperfast":
        st = synth.fast_source(f0, 44100, 512)
        sig = synth.combsubsuperfast_synth(f0, st, c[0], c[1], c[2], c[3], noise, w, 44100, 512)
    else:
        st = synth.phase(f0, 44100, 512)
        fn = synth.sins_synth if kind == "sins" else synth.combsub_synth
        sig = fn(f0, st, c[0], c[1], c[2], noise, 44100, 512)[0]
    (sig * R).sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
print(kind, "forward+backward ms/step", round((time.perf_counter() - t0) / 5 * 1e3, 3))
