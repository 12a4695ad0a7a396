import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.isdir('/root/repo/ddsp_svc_amd') else os.getcwd())
from ddsp_svc_amd import synth
from oracle import ddsp_oracle as O
dev = torch.device("cuda:0")
B, F = 32, 862
f0 = torch.from_numpy(O.synth_f0(B, F, 44100, 512, seed=1)).to(dev)
for _ in range(50): st = synth.phase(f0, 44100, 512)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(2000): st = synth.phase(f0, 44100, 512)
e1.record(); torch.cuda.synchronize()
print("phase call: %.2f us" % (e0.elapsed_time(e1) * 1e3 / 2000))
