#!/usr/bin/env python
"""SURVEY 8-f #2 asks for get_mel "fused onto the synth output".  What a fusion could win: the CombSub step and the log-mel of
its waveform (diffusion/vocoder.py:248: the cascade's next operation) timed apart, back to back (the waveform the last filter
just wrote is still in the 256 MB memory-side cache when k_mel reads it), and with the cache flushed in between (a 512 MB fill:
the waveform comes from HBM) -- steady state, B = 32 x 10 s."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from ddsp_svc_amd import mel as M

dev = torch.device("cuda:0")
B, F = 32, 862
step, inp = bench.build_step("combsub", B, F, 256, dev, seed=1)
stft = M.STFT(44100, 128, 2048, 2048, 512, 40, 16000)
big = torch.empty(128 << 20, dtype=torch.float32, device=dev)


def timeit(fn, reps=300, warm=300):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


y = step()
t_step = timeit(step)
t_mel = timeit(lambda: stft.get_mel(y))
t_pair = timeit(lambda: stft.get_mel(step()))
t_fill = timeit(lambda: big.fill_(0.0), reps=50, warm=20)
t_cold = timeit(lambda: (step(), big.fill_(0.0), stft.get_mel(y)), reps=100, warm=100) - t_fill
print("step %.4f ms | mel alone (waveform cache-resident) %.4f | step + mel back to back %.4f (sum of the two %.4f) | with a 512 MB "
      "fill between them, fill time taken off: %.4f" % (t_step, t_mel, t_pair, t_step + t_mel, t_cold))
