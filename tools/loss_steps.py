#!/usr/bin/env python
"""N steps of the random-scale loss (4 pinned sizes, forward + d/dx_pred) and nothing else: what profilers wrap."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddsp_svc_amd import loss as L

dev = torch.device("cuda:0")
n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B, T = 32, 441344
g = torch.Generator().manual_seed(99)
xt = (torch.randn(B, T, generator=g) * 0.1).to(dev)
xp = (xt * 0.9 + 0.02 * torch.randn(B, T, generator=g).to(dev)).requires_grad_(True)
rss = L.RSSLoss(256, 2048, 4, device=dev)
sizes = torch.tensor([1153, 397, 2011, 768])
real = torch.randint
for _ in range(n_steps):
    torch.randint = lambda *a, **k: sizes
    try:
        value = rss(xp, xt)
    finally:
        torch.randint = real
    grad, = torch.autograd.grad(value, xp)
torch.cuda.synchronize()
print("loss", float(value.detach()))
