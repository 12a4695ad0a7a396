#!/usr/bin/env python
"""forward + backward of the DSP tails (B=32 x 10 s) for rocprofv3 --kernel-trace: which kernels a training step spends
its time in.  `train_step_probe.py <kind> [loss]`: with `loss`, the objective is the reference's RSSLoss(256, 2048, 4)
against a target waveform (train.py:69, solver.py:93-103), its four sizes pinned to one draw (1153, 397, 2011, 768);
without, a plain weighted sum (the synthesiser's own kernels only).  Gradients are taken with torch.autograd.grad: .backward()
would add each step's gradient onto the leaves' .grad (one [B, F, n] add per control), which a training loop -- whose
controls are activations of Unit2Control, not leaves -- never runs."""
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
with_loss = len(sys.argv) > 2 and sys.argv[2] == "loss"
if kind == "combsubsuperfast":
    f0, ctrls, noise = bench.make_inputs(kind, B, F, (1025,) * 4, dev, 1234)
    w = torch.hann_window(2048, device=dev)
elif kind == "combsubfast":
    f0, ctrls, noise = bench.make_inputs(kind, B, F, bench.model_sizes(kind, n), dev, 1234)
    w = torch.sqrt(torch.hann_window(1024, device=dev))
elif kind == "combsub512":                                # the classic configuration: n_mag 256 / 512 / 256 (harmonic filter N = 1022)
    f0, ctrls, noise = bench.make_inputs("combsub", B, F, (256, 512, 256), dev, 1234)
else:
    f0, ctrls, noise = bench.make_inputs(kind, B, F, (n, n, n), dev, 1234)
c = [x.clone().requires_grad_(True) for x in ctrls]
R = torch.randn(B, F * 512, device=dev)
if with_loss:
    from ddsp_svc_amd import loss as hloss
    rss = hloss.RSSLoss(256, 2048, 4, device=dev)
    drawn = torch.tensor([1153, 397, 2011, 768])
    target = R * 0.1


def objective(sig):
    if not with_loss:
        return (sig * R).sum()
    real = torch.randint
    torch.randint = lambda *a, **k: drawn
    try:
        return rss(sig, target)
    finally:
        torch.randint = real


def step():
    if kind == "combsubsuperfast":
        st = synth.fast_source(f0, 44100, 512)
        sig = synth.combsubsuperfast_synth(f0, st, c[0], c[1], c[2], c[3], noise, w, 44100, 512)
    elif kind == "combsubfast":
        st = synth.phase(f0, 44100, 512)
        sig = synth.combsubfast_synth(f0, st, c[0], c[1], c[2], noise, w, 44100, 512)
    else:
        st = synth.phase(f0, 44100, 512)
        fn = synth.sins_synth if kind == "sins" else synth.combsub_synth
        sig = fn(f0, st, c[0], c[1], c[2], noise, 44100, 512)[0]
    return torch.autograd.grad(objective(sig), c)


N_WARM, N_STEPS = int(os.environ.get("TRAIN_WARM", 300)), int(os.environ.get("TRAIN_STEPS", 50))   # the first ~100 steps of a process
for _ in range(N_WARM):                                                                              # are a clock transient
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N_STEPS):
    step()
torch.cuda.synchronize()
print(kind, "forward+backward" + (" with RSSLoss" if with_loss else ""), "ms/step", round((time.perf_counter() - t0) / N_STEPS * 1e3, 3))
