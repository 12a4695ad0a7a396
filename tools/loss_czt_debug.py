"""GPU check of csrc/loss_czt.hip at full size: (1) the loss and d/dx_pred against the eager composition at eps = 1e-7 and
at eps = 1e-3 (well conditioned: no sign(.) flips on bins where S_true ~ S_pred); (2) the backward kernel's transform
alone: the gradient of the spectrum by k_sss_grad, taken back through torch's float64 rfft adjoint."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ddsp_svc_amd import loss as L, _ffi
from ddsp_svc_amd._ffi import ptr

dev = torch.device("cuda:0")
B, T = 32, 441344
g = torch.Generator(device="cpu").manual_seed(5)
xt = (torch.randn(B, T, generator=g) * 0.1).to(dev)
xp = (xt * 0.9 + 0.02 * torch.randn(B, T, generator=g).to(dev)).requires_grad_(True)
lib = _ffi.lib()
for n in (256, 397, 768, 1153, 2047):
    for eps in (1e-7, 1e-3):
        f = L.SSSLoss(n, eps=eps).to(dev)
        loss = f(xt, xp)
        grad, = torch.autograd.grad(loss, xp)
        w = f.spec.window
        sp = lambda x: torch.stft(x, n, hop_length=n, win_length=n, window=w, center=False,
                                  return_complex=True).abs() / w.pow(2).sum().sqrt() + eps
        St, Sp = sp(xt), sp(xp)
        ref = torch.mean(torch.linalg.norm(St - Sp, dim=(1, 2)) / torch.linalg.norm(St + Sp, dim=(1, 2))) \
            + torch.nn.functional.l1_loss(St.log(), Sp.log())
        rgrad, = torch.autograd.grad(ref, xp)
        print(n, eps, "loss rel", abs(float(loss) - float(ref)) / float(ref),
              "grad relrms", float((grad - rgrad).pow(2).mean().sqrt() / rgrad.pow(2).mean().sqrt()), flush=True)
    # (2) transform alone
    f = L.SSSLoss(n).to(dev)
    tab = L._czt_tables(n, xt)
    frames, bins = T // n, n // 2 + 1
    spec = torch.empty(2, B, frames, bins, dtype=torch.complex64, device=dev)
    nb = lib.ddsp_hip_stft_loss_scratch_bytes(B, T, n, n)
    scratch = torch.empty(nb // 8, dtype=torch.float64, device=dev)
    norms = torch.empty(B, 2, device=dev); lo = torch.empty((), device=dev)
    inv = f.spec.inv_window_norm
    _ffi.check(lib.ddsp_hip_stft_loss(ptr(xt), ptr(xp.detach()), B, T, T, n, n, ptr(tab), inv, 1e-7, 1.0, ptr(scratch), nb,
                                      ptr(spec[0]), ptr(spec[1]), ptr(norms), ptr(lo), _ffi.stream_of(xt)))
    X64 = lambda x: torch.fft.rfft(x.double()[:, :frames * n].reshape(B, frames, n) * f.spec.window.double(), dim=-1)
    rt, rp = X64(xt), X64(xp.detach())
    print(n, "spectra: rel err true %.2e pred %.2e" % (float((spec[0] - rt).abs().max() / rt.abs().max()),
                                                       float((spec[1] - rp).abs().max() / rp.abs().max())))
    go = torch.ones((), device=dev)
    G = torch.empty_like(spec[1])
    _ffi.check(lib.ddsp_hip_spectral_loss_backward(ptr(spec[0]), ptr(spec[1]), B, frames * bins, ptr(norms), inv, 1e-7, 1.0,
                                                   ptr(go), 0, ptr(G), _ffi.stream_of(xt)))
    xx = xp.detach().double().requires_grad_(True)
    X64(xx).backward(G.to(torch.complex128))
    d = torch.empty(B, T, device=dev)
    _ffi.check(lib.ddsp_hip_stft_loss_backward(ptr(spec[0]), ptr(spec[1]), B, T, n, n, ptr(tab), ptr(norms), inv, 1e-7, 1.0,
                                               ptr(go), 0, ptr(d), T, 0, None, 0, _ffi.stream_of(xt)))
    e = (d - xx.grad).pow(2).mean().sqrt() / xx.grad.pow(2).mean().sqrt()
    print(n, "backward transform alone: relrms %.2e, worst utterance %.2e" % (
        float(e), float(((d - xx.grad).pow(2).mean(1).sqrt() / xx.grad.pow(2).mean(1).sqrt()).max())), flush=True)
