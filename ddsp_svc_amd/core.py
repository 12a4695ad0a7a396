"""Drop-in for the functions of the reference's ``ddsp/core.py`` that sit on the synthesis path
(same names, positional order and defaults), computed by the HIP kernels of libddsp_hip.so.

Reference lines are cited per function.  All tensors must be float32 (complex64 for responses)
on the GPU; errors mirror the reference (``ValueError`` on batch mismatch / bad padding).
"""
import math
import threading

import torch

from . import _ffi
from ._ffi import ptr

_TABLES = {}
_TABLE_LOCK = threading.Lock()


def _f32c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def ir_table(n_mag, device):
    """Cosine/sine/Hann basis for ``n_mag`` bins, built once per (device, n_mag) on the GPU and kept for the life of the
    process.  Built under a lock and followed by a synchronisation of the building stream (one-time cost), so that callers
    on other host threads / streams never see a half-built or replaced table."""
    key = (str(device), int(n_mag))
    tab = _TABLES.get(key)
    if tab is None:
        L = _ffi.lib()
        with _TABLE_LOCK:
            tab = _TABLES.get(key)
            if tab is None:
                nbytes = L.ddsp_hip_ir_table_bytes(int(n_mag))
                tab = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
                _ffi.check_device(tab)
                _ffi.check(L.ddsp_hip_ir_table(int(n_mag), ptr(tab), _ffi.stream_of(tab)))
                if tab.is_cuda:
                    torch.cuda.current_stream(tab.device).synchronize()
                _TABLES[key] = tab
    return tab


def get_fft_size(frame_size, ir_size, power_of_2=True):
    """core.py:47-63 (kept for API parity; the HIP path has no FFT size)."""
    convolved = ir_size + frame_size - 1
    return int(2 ** math.ceil(math.log2(convolved))) if power_of_2 else convolved


def _upsample_forward(sig, factor):
    B, F, C = sig.shape
    out = torch.empty(B, F * factor, C, dtype=torch.float32, device=sig.device)
    _ffi.check(_ffi.lib().ddsp_hip_upsample(ptr(sig), B, F, C, factor, ptr(out), _ffi.stream_of(sig)))
    return out


class UpsampleFunction(torch.autograd.Function):
    """``upsample`` under autograd (the reference's ``F.interpolate`` is differentiable: amplitudes train through it).
    The adjoint of the interpolation -- frame f collects ``(1 - j/hop)`` of its own block and ``j/hop`` of the previous
    one, the held last frame both halves of the last block -- is two small reductions, left to torch."""

    @staticmethod
    def forward(ctx, signal, factor):
        ctx.factor = factor
        return _upsample_forward(_f32c(signal.detach()), factor)

    @staticmethod
    def backward(ctx, grad_out):
        hop = ctx.factor
        B, T, C = grad_out.shape
        g = grad_out.reshape(B, T // hop, hop, C)
        lam = (torch.arange(hop, device=g.device, dtype=g.dtype) / hop).view(1, 1, hop, 1)
        own, nxt = (g * (1 - lam)).sum(2), (g * lam).sum(2)
        d = own.clone()
        d[:, 1:] += nxt[:, :-1]
        d[:, -1] += nxt[:, -1]                                     # core.py:68: the last frame is repeated
        return d, None


def upsample(signal, factor):
    """core.py:66-70: ``[B,F,C] -> [B,F*factor,C]`` linear interpolation, last frame held.  Differentiable."""
    _ffi.check_device(signal)
    factor = int(factor)
    if torch.is_grad_enabled() and signal.requires_grad:
        return UpsampleFunction.apply(signal, factor)
    return _upsample_forward(_f32c(signal), factor)


def _remove_above_fmax_forward(a, p, fmax, level_start):
    H = a.shape[-1]
    out = torch.empty_like(a)
    _ffi.check(_ffi.lib().ddsp_hip_remove_above_fmax(ptr(a), ptr(p), a.numel() // H, H, float(fmax), int(level_start),
                                                     ptr(out), _ffi.stream_of(a)))
    return out


class RemoveAboveFmaxFunction(torch.autograd.Function):
    """``remove_above_fmax`` under autograd: linear in the amplitudes (the mask, piecewise constant in the pitch, carries
    no gradient -- as in the reference, where it goes through ``.float()`` of a comparison)."""

    @staticmethod
    def forward(ctx, amplitudes, pitch, fmax, level_start):
        a = _f32c(amplitudes.detach())
        p = _f32c(pitch.detach().expand(*a.shape[:-1], 1))
        ctx.save_for_backward(p)
        ctx.cfg = (float(fmax), int(level_start))
        return _remove_above_fmax_forward(a, p, fmax, level_start)

    @staticmethod
    def backward(ctx, grad_out):
        (p,) = ctx.saved_tensors
        return _remove_above_fmax_forward(_f32c(grad_out), p, *ctx.cfg), None, None, None


def remove_above_fmax(amplitudes, pitch, fmax, level_start=1):
    """core.py:73-77: ``amplitudes * ((pitch*k < fmax) + 1e-7)``.  Differentiable w.r.t. the amplitudes."""
    _ffi.check_device(amplitudes, pitch)
    if torch.is_grad_enabled() and amplitudes.requires_grad:
        return RemoveAboveFmaxFunction.apply(amplitudes, pitch, fmax, level_start)
    a = _f32c(amplitudes)
    return _remove_above_fmax_forward(a, _f32c(pitch.expand(*a.shape[:-1], 1)), fmax, level_start)


def crop_and_compensate_delay(audio, audio_size, ir_size, padding="same", delay_compensation=-1):
    """core.py:80-117 (pure slicing; ``fft_convolve`` here already returns the cropped signal)."""
    if padding == "valid":
        crop_size = ir_size + audio_size - 1
    elif padding == "same":
        crop_size = audio_size
    else:
        raise ValueError("Padding must be 'valid' or 'same', instead of {}.".format(padding))
    total = int(audio.shape[-1])
    crop = total - crop_size
    start = ir_size // 2 if delay_compensation < 0 else delay_compensation
    end = crop - start
    return audio[:, start:-end]


def _window_taps(ir, mode, hw):
    """zero-phase taps [..., N] -> windowed causal taps (one elementwise kernel)"""
    x = _f32c(ir)
    N = x.shape[-1]
    rows = x.numel() // N
    out = torch.empty_like(x)
    _ffi.check(_ffi.lib().ddsp_hip_window_impulse_response(ptr(x), int(mode), ptr(hw), rows, N, ptr(out), _ffi.stream_of(x)))
    return out


def _windowed(impulse_response, mode, hw):
    _ffi.check_device(impulse_response, hw)
    if torch.is_grad_enabled() and impulse_response.requires_grad:
        # differentiable composition: the kernel, applied to ones, IS the window in causal order
        w = _window_taps(torch.ones_like(impulse_response, dtype=torch.float32), mode, hw)
        return impulse_response.roll(impulse_response.size(-1) // 2, -1) * w
    return _window_taps(impulse_response, mode, hw)


def apply_window_to_impulse_response(impulse_response, window_size: int = 0, causal: bool = False):
    """core.py:185-237: periodic Hann window on zero-phase taps ``[B,F,N]``, result in causal form.  Every caller of the
    reference uses the defaults (window = the full tap length).  ``causal=True`` fails in the reference itself (it calls
    the non-existent ``torch.fftshift``, core.py:203-204) and a shorter ``window_size`` is a branch no caller reaches;
    both are refused here rather than guessed."""
    if causal:
        raise AttributeError("module 'torch' has no attribute 'fftshift'")          # what the reference raises, core.py:204
    N = int(impulse_response.size(-1))
    if 0 < window_size < N:
        raise NotImplementedError("apply_window_to_impulse_response: window_size < ir_size is not on the synthesis path")
    return _windowed(impulse_response, _ffi.MODE_HANN, None)


def apply_dynamic_window_to_impulse_response(impulse_response, half_width_frames):
    """core.py:240-251: f0-dependent raised-cosine window (``half_width_frames [B,F,1]``; only ``w > 1`` is clamped,
    core.py:245) on zero-phase taps ``[B,F,N]``, result in causal form."""
    hw = _f32c(half_width_frames.detach().expand(*impulse_response.shape[:-1], 1))
    return _windowed(impulse_response, _ffi.MODE_DYNAMIC, hw)


def _impulse_response_forward(re, im, mode, hw):
    B, F, n = re.shape
    taps = torch.empty(B, F, 2 * (n - 1), dtype=torch.float32, device=re.device)
    _ffi.check(_ffi.lib().ddsp_hip_impulse_response(ptr(re), n, ptr(im), n, _ffi.ACT_NONE, 1.0, mode, ptr(hw),
                                                    B * F, n, ptr(ir_table(n, re.device)), ptr(taps),
                                                    _ffi.stream_of(re)))
    return taps


class FrequencyImpulseResponseFunction(torch.autograd.Function):
    """core.py:254-270 with the gradient back to the one-sided response: the adjoint of irfft + roll + window
    (``ddsp_hip_impulse_response_backward``), for real and complex responses in every window mode (the reference
    builds its magnitude responses as ``torch.complex(param, 0)``, vocoder.py:606,849,857); the window itself (half
    widths) is a constant."""

    @staticmethod
    def forward(ctx, magnitudes, mode, hw):
        if magnitudes.is_complex():
            re, im = _f32c(magnitudes.real.detach()), _f32c(magnitudes.imag.detach())
        else:
            re, im = _f32c(magnitudes.detach()), None
        ctx.cfg = (int(mode), im is not None, tuple(re.shape))
        ctx.save_for_backward(hw if hw is not None else re.new_empty(0))
        return _impulse_response_forward(re, im, mode, hw)

    @staticmethod
    def backward(ctx, d_taps):
        (hw,) = ctx.saved_tensors
        mode, is_complex, (B, F, n) = ctx.cfg
        g = _f32c(d_taps)
        d_re = torch.empty(B, F, n, dtype=torch.float32, device=g.device)
        d_im = torch.empty_like(d_re) if is_complex else None
        _ffi.check(_ffi.lib().ddsp_hip_impulse_response_backward(
            ptr(g), None, 0, _ffi.ACT_NONE, 1.0, mode, ptr(hw) if hw.numel() else None, B * F, n,
            ptr(ir_table(n, g.device)), ptr(d_re), ptr(d_im), _ffi.stream_of(g)))
        return (torch.complex(d_re, d_im) if is_complex else d_re), None, None


def frequency_impulse_response(magnitudes, hann_window=True, half_width_frames=None):
    """core.py:254-270: one-sided response ``[B,F,n]`` (complex or real) -> taps ``[B,F,2(n-1)]``
    in causal form, windowed as the flags select.  Differentiable w.r.t. ``magnitudes``."""
    _ffi.check_device(magnitudes, half_width_frames)
    B, F, n = magnitudes.shape
    if not hann_window:
        mode, hw = _ffi.MODE_ROLL, None
    elif half_width_frames is None:
        mode, hw = _ffi.MODE_HANN, None
    else:
        mode, hw = _ffi.MODE_DYNAMIC, _f32c(half_width_frames.detach().expand(B, F, 1))
    if torch.is_grad_enabled() and magnitudes.requires_grad:
        return FrequencyImpulseResponseFunction.apply(magnitudes, mode, hw)
    if magnitudes.is_complex():
        re, im = _f32c(magnitudes.real), _f32c(magnitudes.imag)
    else:
        re, im = _f32c(magnitudes), None
    return _impulse_response_forward(re, im, mode, hw)


def _fft_convolve_forward(x, ir, impl):
    B, T = x.shape
    _, F, N = ir.shape
    hop = T // F
    out = torch.empty(B, T, dtype=torch.float32, device=x.device)
    _ffi.check(_ffi.lib().ddsp_hip_fft_convolve(ptr(x), 0, ptr(ir), None, ptr(out), None, B, F, hop, N, int(impl),
                                                _ffi.stream_of(x)))
    return out


def fft_convolve_backward(grad_out, audio, impulse_response, need_audio_grad=True):
    """Adjoints of ``fft_convolve``: ``(d_audio|None, d_impulse_response)`` for ``grad_out = dL/dout [B,T]``
    (hop 512, N <= 512: the hop-block FFT kernel; other shapes: direct correlations)."""
    x, ir, g = _f32c(audio), _f32c(impulse_response), _f32c(grad_out)
    B, T = x.shape
    _, F, N = ir.shape
    hop = T // F
    d_x = torch.empty_like(x) if need_audio_grad else None
    d_ir = torch.empty_like(ir)
    _ffi.check(_ffi.lib().ddsp_hip_fft_convolve_backward(ptr(x), 0, ptr(ir), ptr(g), ptr(d_x), ptr(d_ir), B, F, hop, N,
                                                         _ffi.stream_of(x)))
    return d_x, d_ir


class FftConvolveFunction(torch.autograd.Function):
    """``fft_convolve`` with autograd through the HIP adjoint kernel (csrc/fir_blk_bwd.hip)."""

    @staticmethod
    def forward(ctx, audio, impulse_response, impl):
        x, ir = _f32c(audio.detach()), _f32c(impulse_response.detach())
        ctx.save_for_backward(x, ir)
        ctx.need_x = audio.requires_grad
        return _fft_convolve_forward(x, ir, impl)

    @staticmethod
    def backward(ctx, grad_out):
        x, ir = ctx.saved_tensors
        d_x, d_ir = fft_convolve_backward(grad_out.contiguous(), x, ir, need_audio_grad=ctx.need_x)
        return d_x, d_ir, None


class FftConvolveAddFunction(torch.autograd.Function):
    """``(fft_convolve(audio, ir) + addend, fft_convolve(audio, ir))`` from ONE launch (the kernel's ``addend`` /
    ``out_plain`` arguments: vocoder.py:609,860 ``signal = harmonic + noise``), differentiable in all three inputs: the
    training composition does not spend an elementwise kernel and a round trip of [B,T] on the sum."""

    @staticmethod
    def forward(ctx, audio, impulse_response, addend, impl):
        x, ir, ad = _f32c(audio.detach()), _f32c(impulse_response.detach()), _f32c(addend.detach())
        ctx.save_for_backward(x, ir)
        ctx.need_x = audio.requires_grad
        ctx.set_materialize_grads(False)
        B, T = x.shape
        _, F, N = ir.shape
        out = torch.empty(B, T, dtype=torch.float32, device=x.device)
        plain = torch.empty_like(out)
        _ffi.check(_ffi.lib().ddsp_hip_fft_convolve(ptr(x), 0, ptr(ir), ptr(ad), ptr(out), ptr(plain), B, F, T // F, N, int(impl),
                                                    _ffi.stream_of(x)))
        return out, plain

    @staticmethod
    def backward(ctx, g_sum, g_plain):
        x, ir = ctx.saved_tensors
        if g_sum is None and g_plain is None:
            return None, None, None, None
        g = g_sum if g_plain is None else (g_plain if g_sum is None else g_sum + g_plain)
        d_x, d_ir = fft_convolve_backward(g.contiguous(), x, ir, need_audio_grad=ctx.need_x)
        return d_x, d_ir, g_sum, None


def fft_convolve_add(audio, impulse_response, addend, impl=_ffi.FIR_AUTO):
    """``(fft_convolve(audio, impulse_response) + addend, fft_convolve(audio, impulse_response))`` in one launch"""
    _ffi.check_device(audio, impulse_response, addend)
    if impulse_response.dim() == 2:
        impulse_response = impulse_response.unsqueeze(1)
    if addend.shape != audio.shape:
        raise ValueError("addend must have the shape of audio")
    if impulse_response.shape[0] != audio.shape[0]:
        raise ValueError("Batch size of audio ({}) and impulse response ({}) must be the same.".format(
            audio.shape[0], impulse_response.shape[0]))
    if audio.shape[1] % impulse_response.shape[1] != 0:
        raise ValueError("audio length {} is not a multiple of the {} impulse-response frames".format(
            audio.shape[1], impulse_response.shape[1]))
    return FftConvolveAddFunction.apply(audio, impulse_response, addend, impl)


def fft_convolve(audio, impulse_response, impl=_ffi.FIR_AUTO):
    """core.py:120-182: time-varying FIR of ``audio [B,T]`` with ``impulse_response [B,F,N]``
    (or ``[B,N]`` for a single filter), ``T = F*hop``; returns ``[B,T]``.  Differentiable w.r.t. both arguments
    (every hop and even N; the fast adjoint kernel at hop 512, N <= 512)."""
    _ffi.check_device(audio, impulse_response)
    if impulse_response.dim() == 2:
        impulse_response = impulse_response.unsqueeze(1)
    Bi, F, N = impulse_response.shape
    B, T = audio.shape
    if B != Bi:
        raise ValueError("Batch size of audio ({}) and impulse response ({}) must be the same.".format(B, Bi))
    hop = int(T / F)
    if hop * F != T:
        raise ValueError("audio length {} is not a multiple of the {} impulse-response frames".format(T, F))
    if torch.is_grad_enabled() and (audio.requires_grad or impulse_response.requires_grad):
        return FftConvolveFunction.apply(audio, impulse_response, impl)
    return _fft_convolve_forward(_f32c(audio), _f32c(impulse_response), impl)


def frequency_filter(audio, magnitudes, hann_window=True, half_width_frames=None):
    """core.py:273-280.  Without gradients: one C call (tap synthesis + time-varying FIR, the taps in a scratch
    tensor); with gradients: the differentiable composition."""
    needs_grad = torch.is_grad_enabled() and (audio.requires_grad or magnitudes.requires_grad)
    if needs_grad or audio.dim() != 2 or magnitudes.dim() != 3:
        return fft_convolve(audio, frequency_impulse_response(magnitudes, hann_window, half_width_frames))
    _ffi.check_device(audio, magnitudes, half_width_frames)
    B, T = audio.shape
    F, n = magnitudes.shape[1], magnitudes.shape[2]
    if magnitudes.shape[0] != B:
        raise ValueError("Batch size of audio ({}) and impulse response ({}) must be the same.".format(
            B, magnitudes.shape[0]))                                           # core.py:151-153
    if F < 1 or T % F != 0 or n < 2:
        return fft_convolve(audio, frequency_impulse_response(magnitudes, hann_window, half_width_frames))
    hop = T // F
    if magnitudes.is_complex():
        re, im = _f32c(magnitudes.real), _f32c(magnitudes.imag)
    else:
        re, im = _f32c(magnitudes), None
    mode = _ffi.MODE_ROLL if not hann_window else (_ffi.MODE_HANN if half_width_frames is None else _ffi.MODE_DYNAMIC)
    hw = None if mode != _ffi.MODE_DYNAMIC else _f32c(half_width_frames.expand(B, F, 1))
    x = _f32c(audio)
    lib = _ffi.lib()
    nbytes = lib.ddsp_hip_frequency_filter_workspace_bytes(B, F, n)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    out = torch.empty(B, T, dtype=torch.float32, device=x.device)
    _ffi.check(lib.ddsp_hip_frequency_filter(ptr(x), ptr(re), n, ptr(im), n, mode, ptr(hw), B, F, hop, n,
                                             ptr(ir_table(n, x.device)), ptr(out), ptr(ws), nbytes, _ffi.stream_of(x)))
    return out
