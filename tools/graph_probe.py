#!/usr/bin/env python
"""Does capturing the step in a hipGraph (torch.cuda.CUDAGraph) change the step time?  (diagnostics)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddsp_svc_amd import synth
import bench

dev = torch.device("cuda:0")
B, F, n = 32, 862, 256
f0, ctrls, noise = bench.make_inputs("combsub", B, F, (n, n, n), dev, 1234)

def step():
    st = synth.phase(f0, 44100, 512)
    return synth.combsub_synth(f0, st, ctrls[0], ctrls[1], ctrls[2], noise, 44100, 512, want_components=False)[0]

def timeit(fn, k=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3

print("eager  ms/step:", timeit(step))
t0 = time.perf_counter()
for _ in range(200): step()
print("cpu-side issue time per step (no sync) ms:", (time.perf_counter() - t0) / 200 * 1e3)
torch.cuda.synchronize()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
print("graph  ms/step:", timeit(g.replay))
