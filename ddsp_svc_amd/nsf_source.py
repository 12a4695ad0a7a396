"""Drop-in for the harmonic source of NSF-HiFiGAN, ``nsf_hifigan.models.SourceModuleHnNSF`` (models.py:174-204, with
``SineGen`` :101-171 inside): ``forward(f0 [B,L], upp) -> [B, L*upp, 1]``, ``tanh(Linear(sine waves * uv + noise))``.

The reference materialises ``[B, L*upp, dim]`` sine waves, noise, voiced mask and noise amplitudes; the HIP kernel
(csrc/sinegen.hip) fuses everything behind the two random draws into one pass.  By default the draws themselves stay
``torch.rand`` / ``torch.randn`` on the device (same distributions, same order as models.py:150,168), so seeding the
torch generator controls them as in the reference.  Opt-in (``in_kernel_noise_seed``): the standard-normal draw -- ``[B, L*upp,
dim]`` floats, 0.5 GB at B = 32 x 10 s and 1.0 GB at BASELINE cfg 5's B = 64 -- happens INSIDE the kernel (a Philox / Box-Muller
stream of its own keyed by (seed, call number); ``normal_noise`` writes the same numbers out), and nothing of that size is
allocated."""
import torch

from . import _ffi
from ._ffi import ptr


def normal_noise(B, T, dim, seed, offset, device):
    """The counter-based standard-normal draw ``z [B, T, dim]`` that ``sine_source(noise=None, noise_seed=..)`` makes inside
    its kernel (``ddsp_hip_normal_noise``; Philox4x32-10 + Box-Muller, a stream of its own -- not torch.randn's)."""
    out = torch.empty(B, T, dim, dtype=torch.float32, device=device)
    _ffi.check_device(out)
    _ffi.check(_ffi.lib().ddsp_hip_normal_noise(int(seed), int(offset), B, T, dim, ptr(out), _ffi.stream_of(out)))
    return out


def sine_source(f0, upp, sampling_rate, weight, bias, rand_ini, noise, sine_amp=0.1, noise_std=0.003,
                voiced_threshold=0.0, noise_seed=None, noise_offset=0):
    """``tanh(Linear(SineGen(f0, upp)))`` with the random draws given: ``rand_ini [dim]`` (entry 0 must be 0),
    ``noise [B, L*upp, dim]`` -- or None: drawn inside the kernel from ``(noise_seed, noise_offset)`` (``normal_noise`` gives
    the same numbers as a tensor) -> ``[B, L*upp]``."""
    _ffi.check_device(f0, weight, bias, rand_ini, noise)
    if f0.dim() != 2:
        raise ValueError("f0 must be [B, L]")
    if noise is None and noise_seed is None:
        raise ValueError("noise=None needs noise_seed (the in-kernel draw is keyed by (noise_seed, noise_offset))")
    B, L = f0.shape
    upp = int(upp)
    dim = weight.numel()
    c = lambda t: t.detach().to(torch.float32).contiguous()
    f0c, w, bb, ri = c(f0), c(weight).reshape(-1), c(bias).reshape(-1), c(rand_ini).reshape(-1)
    if ri.numel() != dim:
        raise ValueError("rand_ini must be [dim]")
    acc = torch.empty(B, L, dtype=torch.float32, device=f0.device)
    out = torch.empty(B, L * upp, dtype=torch.float32, device=f0.device)
    if noise is None:
        _ffi.check(_ffi.lib().ddsp_hip_sine_source_drawn(ptr(f0c), B, L, upp, float(sampling_rate), ptr(ri), int(noise_seed),
                                                         int(noise_offset), ptr(w), ptr(bb), dim, float(sine_amp),
                                                         float(noise_std), float(voiced_threshold), ptr(acc), ptr(out),
                                                         _ffi.stream_of(f0c)))
        return out
    nz = c(noise)
    if nz.numel() != B * L * upp * dim:
        raise ValueError("noise must be [B, L*upp, dim]")
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
        # Opt-in: an integer seed makes the standard-normal draw happen inside the kernel (see the module docstring); every
        # forward call advances the stream (offset = number of calls so far).  None: torch.randn as the reference.
        self.in_kernel_noise_seed = None
        self._noise_calls = 0

    def forward(self, x, upp):
        # In the reference only SineGen runs under no_grad (models.py:141); Linear + tanh are differentiable, so a
        # generator that is being trained updates l_linear.  The fused kernel has no adjoint, so in that case the
        # pre-linear waves are recovered harmonic by harmonic (one-hot weight, zero bias: out = tanh(s_k), |s_k| < 0.2, so
        # atanh is exact to rounding) and Linear + tanh run in torch -- never a silently frozen layer.
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.l_linear.parameters()):
            with torch.no_grad():
                rand_ini, noise, kw = self._draws(x, upp)
                eye = torch.eye(self.dim, device=x.device)
                zero = torch.zeros(1, device=x.device)
                waves = torch.stack([torch.atanh(sine_source(x, upp, self.sampling_rate, eye[k], zero, rand_ini, noise,
                                                             self.sine_amp, self.noise_std, self.voiced_threshold, **kw))
                                     for k in range(self.dim)], -1)
            return self.l_tanh(self.l_linear(waves))
        with torch.no_grad():
            return self._forward(x, upp)

    def _draws(self, x, upp):
        B, L = x.shape
        rand_ini = torch.rand(1, 1, self.dim, device=x.device)                         # models.py:150
        rand_ini[..., 0] = 0                                                            # models.py:151
        if self.in_kernel_noise_seed is not None:                                       # drawn in the kernel: nothing to allocate
            self._noise_calls += 1
            return rand_ini, None, {"noise_seed": int(self.in_kernel_noise_seed), "noise_offset": self._noise_calls - 1}
        noise = torch.randn(B, L * int(upp), self.dim, dtype=torch.float32, device=x.device)   # randn_like, :168
        return rand_ini, noise, {}

    def _forward(self, x, upp):
        rand_ini, noise, kw = self._draws(x, upp)
        out = sine_source(x, upp, self.sampling_rate, self.l_linear.weight, self.l_linear.bias, rand_ini, noise,
                          self.sine_amp, self.noise_std, self.voiced_threshold, **kw)
        return out.unsqueeze(-1)


def patch_reference_source(in_kernel_noise_seed=None):
    """Route ``nsf_hifigan.models.SourceModuleHnNSF.forward`` of an importable reference checkout through the HIP
    kernel for GPU tensors (9 or 1 harmonics); CPU tensors keep the reference code.  ``in_kernel_noise_seed`` (an integer;
    also settable later as an attribute of a module instance, which wins): the standard-normal draw of models.py:168 happens
    inside the kernel and its ``[B, L*upp, dim]`` tensor -- 1.0 GB at cfg 5's B = 64 x 10 s -- is never allocated."""
    import nsf_hifigan.models as nm
    nm.SourceModuleHnNSF.in_kernel_noise_seed = in_kernel_noise_seed
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
            seed = getattr(self, "in_kernel_noise_seed", None)
            if seed is not None:
                calls = self.__dict__.get("_noise_calls", 0)
                self.__dict__["_noise_calls"] = calls + 1
                noise, kw = None, {"noise_seed": int(seed), "noise_offset": calls}
            else:
                noise, kw = torch.randn(B, L * int(upp), gen.dim, dtype=torch.float32, device=x.device), {}
            out = sine_source(x, upp, gen.sampling_rate, self.l_linear.weight, self.l_linear.bias, rand_ini, noise,
                              gen.sine_amp, gen.noise_std, gen.voiced_threshold, **kw)
        return out.unsqueeze(-1)

    nm.SourceModuleHnNSF.forward = forward
    return nm
