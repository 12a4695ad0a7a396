"""Drop-in for the harmonic source of NSF-HiFiGAN, ``nsf_hifigan.models.SourceModuleHnNSF`` (models.py:174-204, with
``SineGen`` :101-171 inside): ``forward(f0 [B,L], upp) -> [B, L*upp, 1]``, ``tanh(Linear(sine waves * uv + noise))``.

The reference materialises ``[B, L*upp, dim]`` sine waves, noise, voiced mask and noise amplitudes; the HIP kernel
(csrc/sinegen.hip) fuses everything behind the two random draws into one pass.  The draws themselves stay
``torch.rand`` / ``torch.randn`` on the device (same distributions, same order as models.py:150,168), so seeding the
torch generator controls them as in the reference."""
import torch

from . import _ffi
from ._ffi import ptr


def sine_source(f0, upp, sampling_rate, weight, bias, rand_ini, noise, sine_amp=0.1, noise_std=0.003,
                voiced_threshold=0.0):
    """``tanh(Linear(SineGen(f0, upp)))`` with the random draws given: ``rand_ini [dim]`` (entry 0 must be 0),
    ``noise [B, L*upp, dim]`` -> ``[B, L*upp]``."""
    _ffi.check_device(f0, weight, bias, rand_ini, noise)
    if f0.dim() != 2:
        raise ValueError("f0 must be [B, L]")
    B, L = f0.shape
    upp = int(upp)
    dim = weight.numel()
    c = lambda t: t.detach().to(torch.float32).contiguous()
    f0c, w, bb, ri, nz = c(f0), c(weight).reshape(-1), c(bias).reshape(-1), c(rand_ini).reshape(-1), c(noise)
    if nz.numel() != B * L * upp * dim or ri.numel() != dim:
        raise ValueError("noise must be [B, L*upp, dim] and rand_ini [dim]")
    acc = torch.empty(B, L, dtype=torch.float32, device=f0.device)
    out = torch.empty(B, L * upp, dtype=torch.float32, device=f0.device)
    _ffi.check(_ffi.lib().ddsp_hip_sine_source(ptr(f0c), B, L, upp, float(sampling_rate), ptr(ri), ptr(nz), ptr(w), ptr(bb),
                                               dim, float(sine_amp), float(noise_std), float(voiced_threshold), ptr(acc),
                                               ptr(out), _ffi.stream_of(f0c)))
    return out


class SourceModuleHnNSF(torch.nn.Module):
    """nsf_hifigan/models.py:174-204: same constructor, same parameters (``l_linear.weight`` / ``l_linear.bias``), so
    the reference vocoder's checkpoints load unchanged."""

    def __init__(self, sampling_rate, harmonic_num=0, sine_amp=0.1, add_noise_std=0.003, voiced_threshod=0):
        super().__init__()
        self.sampling_rate = sampling_rate
        self.sine_amp = sine_amp
        self.noise_std = add_noise_std
        self.voiced_threshold = voiced_threshod
        self.dim = harmonic_num + 1
        self.l_linear = torch.nn.Linear(harmonic_num + 1, 1)
        self.l_tanh = torch.nn.Tanh()

    def forward(self, x, upp):
        # In the reference only SineGen runs under no_grad (models.py:141); Linear + tanh are differentiable, so a
        # generator that is being trained updates l_linear.  The fused kernel has no adjoint, so in that case the
        # pre-linear waves are recovered harmonic by harmonic (one-hot weight, zero bias: out = tanh(s_k), |s_k| < 0.2, so
        # atanh is exact to rounding) and Linear + tanh run in torch -- never a silently frozen layer.
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.l_linear.parameters()):
            with torch.no_grad():
                rand_ini, noise = self._draws(x, upp)
                eye = torch.eye(self.dim, device=x.device)
                zero = torch.zeros(1, device=x.device)
                waves = torch.stack([torch.atanh(sine_source(x, upp, self.sampling_rate, eye[k], zero, rand_ini, noise,
                                                             self.sine_amp, self.noise_std, self.voiced_threshold))
                                     for k in range(self.dim)], -1)
            return self.l_tanh(self.l_linear(waves))
        with torch.no_grad():
            return self._forward(x, upp)

    def _draws(self, x, upp):
        B, L = x.shape
        rand_ini = torch.rand(1, 1, self.dim, device=x.device)                         # models.py:150
        rand_ini[..., 0] = 0                                                            # models.py:151
        noise = torch.randn(B, L * int(upp), self.dim, dtype=torch.float32, device=x.device)   # randn_like, :168
        return rand_ini, noise

    def _forward(self, x, upp):
        rand_ini, noise = self._draws(x, upp)
        out = sine_source(x, upp, self.sampling_rate, self.l_linear.weight, self.l_linear.bias, rand_ini, noise,
                          self.sine_amp, self.noise_std, self.voiced_threshold)
        return out.unsqueeze(-1)


def patch_reference_source():
    """Route ``nsf_hifigan.models.SourceModuleHnNSF.forward`` of an importable reference checkout through the HIP
    kernel for GPU tensors (9 or 1 harmonics); CPU tensors keep the reference code."""
    import nsf_hifigan.models as nm
    if hasattr(nm.SourceModuleHnNSF, "_reference_forward"):
        return nm
    ref_forward = nm.SourceModuleHnNSF.forward
    nm.SourceModuleHnNSF._reference_forward = ref_forward

    def forward(self, x, upp):
        gen = self.l_sin_gen
        training = torch.is_grad_enabled() and any(p.requires_grad for p in self.l_linear.parameters())
        if not getattr(x, "is_cuda", False) or gen.dim not in (1, 9) or x.dim() != 2 or training:
            return ref_forward(self, x, upp)            # incl. training of l_linear: the reference's differentiable code
        with torch.no_grad():
            B, L = x.shape
            rand_ini = torch.rand(1, 1, gen.dim, device=x.device)
            rand_ini[..., 0] = 0
            noise = torch.randn(B, L * int(upp), gen.dim, dtype=torch.float32, device=x.device)
            out = sine_source(x, upp, gen.sampling_rate, self.l_linear.weight, self.l_linear.bias, rand_ini, noise,
                              gen.sine_amp, gen.noise_std, gen.voiced_threshold)
        return out.unsqueeze(-1)

    nm.SourceModuleHnNSF.forward = forward
    return nm
