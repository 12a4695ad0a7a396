#!/usr/bin/env python
"""Every kernel of the library is deterministic: the same launch on the same inputs must give the same bits.  This probe
repeats each operation at full size many times (other work in between shifts the waves' relative timing) and compares
every result with the first one bit for bit -- a difference is a race between waves or workgroups.  It found the missing
barrier of k_fir_fft (two wrong samples in about one launch of a hundred).  Exit status 1 on any mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ddsp_svc_amd import _ffi, core, synth

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
SR, HOP, n = 44100, 512, 256
B, F = 32, 862
N = 2 * (n - 1)
f0, (c0, c1, c2), noise = bench.make_inputs("combsub", B, F, (n, n, n), dev, 7)
g = torch.Generator().manual_seed(11)
x = (torch.rand(B, F * HOP, generator=g) * 2 - 1).to(dev)
taps = (torch.randn(B, F, N, generator=g) / N ** 0.5).to(dev)
gout = torch.randn(B, F * HOP, generator=g).to(dev)
st = synth.phase(f0, SR, HOP)
fs = synth.fast_source(f0, SR, HOP, want_combtooth=True)
ops = {
    "fft_convolve impl 3 (direct form, MFMA)": lambda: core.fft_convolve(x, taps, impl=3),
    "fft_convolve impl 4 (per-frame FFT form)": lambda: core.fft_convolve(x, taps, impl=4),
    "fft_convolve impl 5 (hop-block FFT form)": lambda: core.fft_convolve(x, taps, impl=5),
    "fft_convolve backward": lambda: torch.cat([t.reshape(-1) for t in core.fft_convolve_backward(gout, x, taps)]),
    "phase": lambda: synth.phase(f0, SR, HOP, want_x=True).x,
    "combtooth": lambda: synth.combtooth(f0, st, SR, HOP),
    "combsub tail (two streams)": lambda: synth.combsub_synth(f0, st, c0, c1, c2, noise, SR, HOP, want_components=False)[0],
    "sins tail (two streams)": lambda: synth.sins_synth(f0, st, c0, c1, c2, noise, SR, HOP, want_components=False)[0],
    "tap synthesis, dynamic window": lambda: core.frequency_impulse_response(torch.exp(c1), half_width_frames=(1.5 * SR / (f0.reshape(B, F, 1) + 1e-3))),
    "fast source": lambda: synth.fast_source(f0, SR, HOP, want_combtooth=True).combtooth,
}
for win, name in ((1024, "combsubfast"), (2048, "combsubsuperfast")):
    nb = win // 2 + 1
    gg = torch.Generator().manual_seed(win)
    hm, hp, nm, nph = (torch.randn(B, F, nb, generator=gg).mul_(0.3).to(dev) for _ in range(4))
    w = torch.hann_window(win).to(dev) if win == 2048 else torch.sqrt(torch.hann_window(win)).to(dev)
    gz = torch.randn(B, F * HOP, generator=gg).to(dev)
    if win == 2048:
        ops["combsubsuperfast tail"] = lambda hm=hm, hp=hp, nm=nm, nph=nph, gz=gz, w=w: synth.combsubsuperfast_synth(f0, fs, hm, hp, nm, nph, gz, w, SR, HOP)
        ops["stft filter backward (win 2048)"] = lambda hm=hm, hp=hp, nm=nm, nph=nph, gz=gz, w=w: torch.cat([t.reshape(-1) for t in synth.stft_filter_backward(gout, fs.combtooth, gz, hm, hp, nm, nph, w, HOP) if t is not None])
    else:
        ops["combsubfast tail"] = lambda hm=hm, hp=hp, nm=nm, w=w: synth.combsubfast_synth(f0, st, hm, hp, nm, noise, w, SR, HOP)
# the remaining operations: autograd through the tails, the mel front-end, the NSF source, the spectral loss
from ddsp_svc_amd import mel as hmel, nsf_source, loss as hloss


def tail_grads(fn, ctrls):
    leaves = [c.detach().clone().requires_grad_(True) for c in ctrls]
    out = fn(f0, st, *leaves, noise, SR, HOP, want_components=False)[0]
    out.backward(gout)
    return torch.cat([l.grad.reshape(-1) for l in leaves])


ops["combsub tail, backward to the controls"] = lambda: tail_grads(synth.combsub_synth, (c0, c1, c2))
ops["sins tail, backward to the controls"] = lambda: tail_grads(synth.sins_synth, (c0, c1, c2))
# paths off the headline shape: dense tap synthesis (other bin counts) and its adjoint, the first sinusoid bank (other hops),
# the simple per-sample filter
n2 = 129
z2 = torch.complex(torch.randn(B, F, n2, generator=g), torch.randn(B, F, n2, generator=g)).to(dev)
ops["tap synthesis, dense contraction (129 bins, complex)"] = lambda: core.frequency_impulse_response(z2)


def taps_grad():
    zz = torch.randn(B, F, n2, generator=torch.Generator().manual_seed(1)).to(dev).requires_grad_(True)
    core.frequency_impulse_response(torch.exp(zz)).square().sum().backward()
    return zz.grad


ops["tap synthesis adjoint (129 bins)"] = taps_grad
f0h = bench.make_inputs("combsub", 8, 300, (64, 64, 64), dev, 3)
sth = synth.phase(f0h[0], SR, 256)
ops["sinusoid bank, hop 256"] = lambda: synth.sinusoid_bank(f0h[0], sth, f0h[1][0], SR, 256)
xs, ts = x[:2, : 40 * 441].contiguous(), torch.randn(2, 40, 64, generator=g).to(dev)
ops["fft_convolve, hop 441 (simple / direct forms)"] = lambda: core.fft_convolve(xs, ts)
ops["fft_convolve, 2u-1 + addend + plain output"] = lambda: torch.cat([t.reshape(-1) for t in (lambda o, p: (_ffi.check(_ffi.lib().ddsp_hip_fft_convolve(
    ((x + 1) / 2).data_ptr(), 1, taps.data_ptr(), gout.data_ptr(), o.data_ptr(), p.data_ptr(), B, F, HOP, N, 5, _ffi.stream_of(x))), (o, p))[1])(torch.empty_like(x), torch.empty_like(x))])
ops["combsub tail, noise drawn in the kernel"] = lambda: synth.combsub_synth(f0, st, c0, c1, c2, None, SR, HOP, want_components=False, noise_seed=5, noise_offset=9)[0]
try:
    basis = torch.as_tensor(hmel.slaney_mel_filterbank(44100, 2048, 128, 40, 16000)).float().to(dev)
    wmel = torch.hann_window(2048).to(dev)
    band = hmel._bands(basis)
    ops["log-mel front-end"] = lambda: hmel.mel_spectrogram(x[:, : 430 * 512], wmel, basis, band, 512)
except Exception as e:                                  # the banded basis helper wants a real filterbank shape
    print("mel skipped:", e)
L_, upp = 862, 512
rnd = torch.rand(9, generator=g).to(dev); rnd[0] = 0
nzs = torch.randn(8, L_ * upp, 9, generator=g).to(dev)
wgt = torch.randn(9, generator=g).to(dev); bia = torch.randn(1, generator=g).to(dev)
ops["NSF harmonic source"] = lambda: nsf_source.sine_source(f0[:8].reshape(8, -1), upp, SR, wgt, bia, rnd, nzs)
lossm = hloss.SSSLoss(1024, 1.0, 0.75).to(dev)


def loss_grad():
    xp = x[:8].detach().clone().requires_grad_(True)
    l = lossm(gout[:8] * 0.1, xp)
    l.backward()
    return torch.cat([l.detach().reshape(-1), xp.grad.reshape(-1)])


ops["spectral loss, forward + backward (ATen stft backward: not deterministic, informational)"] = loss_grad
rssm = hloss.RSSLoss(256, 2048, 4, device=dev)
rss_sizes = torch.tensor([1153, 397, 2011, 768])


def rss_grad():
    real = torch.randint
    torch.randint = lambda *a, **k: rss_sizes
    try:
        xp = x.detach().clone().requires_grad_(True)
        l = rssm(xp, gout * 0.1)
    finally:
        torch.randint = real
    l.backward()
    return torch.cat([l.detach().reshape(-1), xp.grad.reshape(-1)])


ops["random-scale loss from the waveforms (in-kernel chirp-z STFT, 4 sizes), forward + backward"] = rss_grad
bad = 0
filler = torch.empty(64 << 20, device=dev)
for name, fn in ops.items():
    first = fn().clone()
    miss = 0
    for it in range(n_rep):
        if it % 3 == 1:
            filler.normal_()                       # something else on the chip in between
        r = fn()
        if not torch.equal(r, first):
            miss += 1
            d = (r.float() - first.float()).abs()
            if miss <= 3:
                print("  %s: repetition %d differs in %d elements, max |diff| %.3g" % (name, it, int((d > 0).sum()), float(d.max())))
    torch.cuda.synchronize()
    print("%-44s %d repetitions, %d mismatches" % (name, n_rep, miss))
    if "informational" not in name:
        bad += miss
sys.exit(1 if bad else 0)
