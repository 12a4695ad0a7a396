"""Functional layer over the fused C-ABI entry points: phase state, exciters and the two DSP tails
(reference: ddsp/vocoder.py:564-611 Sins, :819-862 CombSub).  Tensors in, tensors out, all on
the GPU; controls may be non-contiguous ``torch.split`` views (row stride is passed through)."""
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _ffi
from ._ffi import ptr
from .core import _f32c, ir_table

# same-box A/B switch (tools/train_step_probe.py): 1 = the noise branch of the training composition is joined into the
# caller's stream as soon as it is launched (the order before round 4) instead of at its first consumer
_EARLY_JOIN = os.environ.get("DDSP_HIP_TRAIN_EARLY_JOIN", "0") == "1"
# same-box A/B switch: 1 = training takes the per-operator composition of rounds 2 - 5 even where the fused tail applies
_TRAIN_COMPOSED = os.environ.get("DDSP_HIP_TRAIN_COMPOSED", "0") == "1"


@dataclass
class PhaseState:
    """Output of HOT-1.  ``phase_frames [B,F,1]`` goes to Unit2Control; ``phase0 [B,F]`` (float64,
    unwrapped cycles accumulated before each frame) is what the exciters restart from."""
    phase0: torch.Tensor
    phase_frames: torch.Tensor
    initial_phase: Optional[torch.Tensor]
    infer: bool
    x: Optional[torch.Tensor] = None


def _rows(t, n):
    """pointer-compatible view of a ``[B,F,n]`` control: last dim contiguous, uniform frame stride"""
    if t.dtype != torch.float32:
        t = t.float()
    B, F, nn = t.shape
    assert nn == n
    if t.stride(2) != 1 or (B > 1 and t.stride(0) != F * t.stride(1)) or t.stride(1) < n:
        t = t.contiguous()
    return t, t.stride(1)


def _initial_phase(initial_phase, B, device):
    if initial_phase is None:
        return None
    ip = initial_phase.to(device=device, dtype=torch.float32).reshape(-1)
    if ip.numel() == 1:
        ip = ip.expand(B)
    if ip.numel() != B:
        raise ValueError("initial_phase must hold one value per utterance")
    return ip.contiguous()


def phase(f0_frames, sampling_rate, block_size, initial_phase=None, infer=True, want_x=False) -> PhaseState:
    """vocoder.py:564-575: upsample f0, cumulative phase (float64 scan if ``infer``), wrap,
    ``phase_frames = 2*pi*x[:, ::block_size]``."""
    _ffi.check_device(f0_frames)
    f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
    B, F = f0.shape
    hop = int(block_size)
    dev = f0.device
    ip = _initial_phase(initial_phase, B, dev)
    sums = torch.empty(B, F, dtype=torch.float64, device=dev)
    phase0 = torch.empty(B, F, dtype=torch.float64, device=dev)
    pf = torch.empty(B, F, 1, dtype=torch.float32, device=dev)
    x = torch.empty(B, F * hop, dtype=torch.float32, device=dev) if want_x else None
    _ffi.check(_ffi.lib().ddsp_hip_phase(ptr(f0), ptr(ip), B, F, hop, float(sampling_rate), int(bool(infer)),
                                         ptr(sums), ptr(phase0), ptr(pf), ptr(x), _ffi.stream_of(f0)))
    return PhaseState(phase0, pf, ip, bool(infer), x)


def combtooth(f0_frames, state: PhaseState, sampling_rate, block_size):
    """vocoder.py:839-840: ``sinc(sr * x / (f0 + 1e-3))`` -> ``[B,T]``."""
    f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
    B, F = f0.shape
    hop = int(block_size)
    out = torch.empty(B, F * hop, dtype=torch.float32, device=f0.device)
    _ffi.check(_ffi.lib().ddsp_hip_combtooth(ptr(f0), ptr(state.initial_phase), ptr(state.phase0), B, F, hop,
                                             float(sampling_rate), int(state.infer), ptr(out), _ffi.stream_of(f0)))
    return out


def sinusoid_bank(f0_frames, state: PhaseState, amplitudes_ctrl, sampling_rate, block_size):
    """vocoder.py:580,585-594 from the raw ``amplitudes`` control -> ``[B,T]``."""
    f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
    B, F = f0.shape
    hop = int(block_size)
    H = amplitudes_ctrl.shape[-1]
    c, ld = _rows(amplitudes_ctrl, H)
    out = torch.empty(B, F * hop, dtype=torch.float32, device=f0.device)
    _ffi.check(_ffi.lib().ddsp_hip_sinusoid_bank(ptr(f0), ptr(state.initial_phase), ptr(state.phase0), ptr(c), ld,
                                                 B, F, hop, H, float(sampling_rate), int(state.infer), ptr(out),
                                                 _ffi.stream_of(f0)))
    return out


def _workspace(B, F, hop, n_max, device):
    """Scratch of one call, taken from torch's caching allocator on the caller's current stream: the allocator's
    stream-ordered reuse rules then cover concurrent callers (two host threads, two torch streams on one device -- the
    GUI calls the model from its audio callback thread, gui.py:393) without any state in this module.  The part the
    noise branch touches on the second stream needs no ``record_stream``: the call joins that stream back into the
    caller's before its last kernel, so everything is ordered before whatever the caller's stream does next."""
    need = _ffi.lib().ddsp_hip_synth_workspace_bytes(B, F, hop, n_max)
    return torch.empty(need, dtype=torch.uint8, device=device), need


def _outputs(B, T, device, want_components):
    signal = torch.empty(B, T, dtype=torch.float32, device=device)
    if want_components:
        return signal, torch.empty_like(signal), torch.empty_like(signal)
    return signal, None, None


def uniform_noise(B, T, seed, offset, device):
    """The counter-based uniform draw ``u [B,T]`` in [0,1) that the tails generate inside their noise filter when
    ``noise=None`` (``ddsp_hip_uniform_noise``; a Philox4x32-10 stream of its own, not torch.rand's)."""
    out = torch.empty(B, T, dtype=torch.float32, device=device)
    _ffi.check_device(out)
    _ffi.check(_ffi.lib().ddsp_hip_uniform_noise(int(seed), int(offset), B, T, ptr(out), _ffi.stream_of(out)))
    return out


def _need_seed(noise, noise_seed):
    """``noise=None`` asks for the counter-based draw, which is keyed by ``(noise_seed, noise_offset)``: checked once, at the
    top of the tails, so that every path (inference, training composition) raises the same documented error"""
    if noise is None and noise_seed is None:
        raise ValueError("noise=None needs noise_seed (the in-kernel draw is keyed by (noise_seed, noise_offset))")


def _noise_arg(noise, noise_seed, noise_offset, noise_is_u01, B, T, hop, n_nz, fir_impl, device):
    """(tensor | None, is_u01, seed, offset) for the C call: ``noise=None`` asks for the in-kernel draw from
    ``(noise_seed, noise_offset)``; where the noise filter's shape is outside the kernel that can draw (hop 512,
    n_mag_noise <= 257) the same numbers are written out first."""
    if noise is not None:
        return _f32c(noise.reshape(B, T)), noise_is_u01, 0, 0
    _need_seed(noise, noise_seed)
    # (the hop-block filter takes utterances below 2^28 samples: launch_fir_blk; longer ones get the same numbers written out)
    if hop == 512 and n_nz <= 257 and T < 2 ** 28 and fir_impl in (_ffi.FIR_AUTO, _ffi.FIR_BLK):
        return None, False, int(noise_seed), int(noise_offset)
    return uniform_noise(B, T, noise_seed, noise_offset, device), True, 0, 0


def sins_synth(f0_frames, state: PhaseState, amplitudes, group_delay, noise_magnitude, noise, sampling_rate,
               block_size, noise_is_u01=False, want_components=True, fir_impl=_ffi.FIR_AUTO, noise_seed=None,
               noise_offset=0):
    """DSP tail of ``Sins.forward`` (vocoder.py:580-611) from raw controls.  ``noise [B,T]`` is the
    uniform draw (``noise_is_u01``: raw ``rand_like`` output, else already ``2u-1``), or None: drawn inside the noise
    filter from ``(noise_seed, noise_offset)`` (``uniform_noise`` gives the same numbers as a tensor).
    Returns ``(signal, harmonic|None, noise|None)``.  With gradients enabled and a control that requires grad the
    differentiable composition is used (the filters at every hop and bin count; their fast adjoint kernels at hop 512, n_mag <= 257)."""
    _ffi.check_device(f0_frames, amplitudes, group_delay, noise_magnitude, noise, state.phase0)
    _need_seed(noise, noise_seed)
    if noise is None and torch.is_grad_enabled() and any(c.requires_grad for c in (amplitudes, group_delay, noise_magnitude)):
        noise, noise_is_u01 = uniform_noise(f0_frames.shape[0], f0_frames.shape[1] * int(block_size), noise_seed, noise_offset,
                                            f0_frames.device), True
    if torch.is_grad_enabled() and any(c.requires_grad for c in (amplitudes, group_delay, noise_magnitude)):
        f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
        B, F = f0.shape
        n = group_delay.shape[-1]
        lay = None
        if _fused_train_ok(f0_frames, block_size, amplitudes, group_delay, noise_magnitude) and fir_impl == _ffi.FIR_AUTO \
                and noise_magnitude.shape[-1] == n and not _TRAIN_COMPOSED:
            lay = _tail_layout(False, B, F, int(block_size), amplitudes.shape[-1], n, n)
        if lay is not None:
            return SinsTailFunction.apply(f0, state, amplitudes, group_delay, noise_magnitude, _f32c(noise.reshape(B, -1)),
                                          noise_is_u01, sampling_rate, int(block_size), lay)
        return _sins_synth_train(f0_frames, state, amplitudes, group_delay, noise_magnitude, noise, sampling_rate,
                                 block_size, noise_is_u01)
    f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
    B, F = f0.shape
    hop = int(block_size)
    T = F * hop
    H, n_ap, n_nz = amplitudes.shape[-1], group_delay.shape[-1], noise_magnitude.shape[-1]
    ca, lda = _rows(amplitudes, H)
    cg, ldg = _rows(group_delay, n_ap)
    cn, ldn = _rows(noise_magnitude, n_nz)
    dev = f0.device
    nz, is_u01, seed, offset = _noise_arg(noise, noise_seed, noise_offset, noise_is_u01, B, T, hop, n_nz, fir_impl, dev)
    ws, need = _workspace(B, F, hop, max(n_ap, n_nz), dev)
    signal, harm, nzo = _outputs(B, T, dev, want_components)
    _ffi.check(_ffi.lib().ddsp_hip_sins_synth(
        ptr(f0), ptr(state.initial_phase), ptr(state.phase0), ptr(ca), lda, ptr(cg), ldg, ptr(cn), ldn,
        ptr(nz), int(is_u01), B, F, hop, float(sampling_rate), int(state.infer), H, n_ap, n_nz,
        ptr(ir_table(n_ap, dev)), ptr(ir_table(n_nz, dev)), ptr(signal), ptr(harm), ptr(nzo),
        ptr(ws), need, int(fir_impl), _ffi.stream_of(f0), _ffi.aux_stream_of(f0, B * F), seed, offset))
    return signal, harm, nzo


class SinusoidBankFunction(torch.autograd.Function):
    """``sinusoid_bank`` with the gradient back to the raw ``amplitudes`` control (every hop the forward takes: the
    matrix-pipe adjoint at hop 512, one wave per frame with direct sines elsewhere)."""

    @staticmethod
    def forward(ctx, f0_frames, state, amplitudes_ctrl, sampling_rate, block_size):
        out = sinusoid_bank(f0_frames, state, amplitudes_ctrl.detach(), sampling_rate, block_size)
        ctx.save_for_backward(_f32c(f0_frames.reshape(f0_frames.shape[0], -1)), amplitudes_ctrl.detach())
        ctx.cfg = (state, float(sampling_rate), int(block_size))
        return out

    @staticmethod
    def backward(ctx, grad_out):
        f0, c_amp = ctx.saved_tensors
        state, sr, hop = ctx.cfg
        B, F = f0.shape
        H = c_amp.shape[-1]
        c, ld = _rows(c_amp, H)
        g = _f32c(grad_out.reshape(B, F * hop))
        lib = _ffi.lib()
        scratch = torch.empty(lib.ddsp_hip_sinusoid_bank_backward_scratch_bytes(B, F, H), dtype=torch.uint8, device=f0.device)
        d_c = torch.empty(B, F, H, dtype=torch.float32, device=f0.device)
        _ffi.check(lib.ddsp_hip_sinusoid_bank_backward(ptr(f0), ptr(state.initial_phase), ptr(state.phase0), ptr(c), ld,
                                                       ptr(g), B, F, hop, H, sr, int(state.infer), ptr(scratch), ptr(d_c),
                                                       _ffi.stream_of(f0)))
        return None, None, d_c, None, None


# ---- differentiable tap synthesis from raw controls (training path of Sins / CombSub) ------------------------
def _rows2(t, n):
    """[B,F,n] control -> (tensor usable as [B*F, n] rows, row stride)"""
    t, ld = _rows(t, n)
    return t, ld


class MagnitudeTapsFunction(torch.autograd.Function):
    """taps = window(roll(irfft(scale * exp(c)))) (vocoder.py:835-836 + core.py:254-270) with the gradient back to
    the raw control ``c [B,F,n]``; ``mode`` HANN or DYNAMIC (``half_width [B,F]``)."""

    @staticmethod
    def forward(ctx, c, scale, mode, half_width):
        B, F, n = c.shape
        cc, ld = _rows(c.detach(), n)
        N = 2 * (n - 1)
        dev = c.device
        taps = torch.empty(B, F, N, dtype=torch.float32, device=dev)
        hw = None if half_width is None else _f32c(half_width.reshape(B * F))
        _ffi.check(_ffi.lib().ddsp_hip_impulse_response(ptr(cc), ld, None, 0, _ffi.ACT_EXP, float(scale), int(mode),
                                                        ptr(hw), B * F, n, ptr(ir_table(n, dev)), ptr(taps),
                                                        _ffi.stream_of(cc)))
        ctx.save_for_backward(cc, hw if hw is not None else cc.new_empty(0))
        ctx.cfg = (ld, float(scale), int(mode), hw is not None)
        return taps

    @staticmethod
    def backward(ctx, d_taps):
        cc, hw = ctx.saved_tensors
        ld, scale, mode, has_hw = ctx.cfg
        B, F, n = cc.shape
        d_c = torch.empty(B, F, n, dtype=torch.float32, device=cc.device)
        g = _f32c(d_taps)
        _ffi.check(_ffi.lib().ddsp_hip_impulse_response_backward(ptr(g), ptr(cc), ld, _ffi.ACT_EXP, scale, mode,
                                                                 ptr(hw) if has_hw else None, B * F, n,
                                                                 ptr(ir_table(n, cc.device)), ptr(d_c), None,
                                                                 _ffi.stream_of(cc)))
        return d_c, None, None, None


class AllpassTapsFunction(torch.autograd.Function):
    """taps = roll(irfft(exp(1j * cumsum(pi * tanh(c))))) (vocoder.py:834,845 + core.py:254-270, no window) with the
    gradient back to the raw group-delay control ``c [B,F,n]``."""

    @staticmethod
    def forward(ctx, c):
        B, F, n = c.shape
        cc, ld = _rows(c.detach(), n)
        dev = c.device
        N = 2 * (n - 1)
        taps = torch.empty(B, F, N, dtype=torch.float32, device=dev)
        lib = _ffi.lib()
        nbytes = lib.ddsp_hip_allpass_taps_scratch_bytes(B * F, n)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _ffi.check(lib.ddsp_hip_allpass_taps(ptr(cc), ld, B * F, n, ptr(ir_table(n, dev)), ptr(taps), ptr(scratch), nbytes,
                                             _ffi.stream_of(cc)))
        ctx.save_for_backward(cc)
        ctx.ld = ld
        return taps

    @staticmethod
    def backward(ctx, d_taps):
        (cc,) = ctx.saved_tensors
        B, F, n = cc.shape
        return _allpass_taps_bwd(_f32c(d_taps).view(B, F, 2 * (n - 1)), cc, ctx.ld)


# ---- training through the FUSED tails (round 6): the forward pass of a training step is the inference call itself -------
# solver.py:93-103 runs the same forward with gradients.  Rounds 2 - 5 composed it from per-operator autograd Functions
# (exciter, three tap syntheses, three filters: seven launches on two streams); the fused entry points leave every intermediate
# the adjoints need in their workspace (ddsp_hip_tail_layout), so the forward of a training step is the three- / four-launch
# inference call and only the backward pass is a sequence of adjoint kernels.

def _tail_layout(combsub, B, F, hop, n0, n1, n2):
    """byte offsets of the intermediates a fused tail call of this shape leaves in its workspace, or None (not the fused layout)"""
    import ctypes
    off = (ctypes.c_longlong * 6)()
    rc = _ffi.lib().ddsp_hip_tail_layout(int(combsub), B, F, hop, n0, n1, n2, 0, 0, ctypes.addressof(off))
    if rc < 0:
        _ffi.check(rc)
    return list(off) if rc == 1 else None


def _ws_view(ws, offset, shape):
    n = 1
    for d in shape:
        n *= d
    return ws[offset:offset + 4 * n].view(torch.float32).view(*shape)


def _fir_bwd(x, x_is_u01, taps, grad, need_dx):
    """(d_x | None, d_taps) of fft_convolve for the cotangent ``grad [B,T]`` (ddsp_hip_fft_convolve_backward)"""
    B, F, N = taps.shape
    T = x.shape[1]
    d_taps = torch.empty(B, F, N, dtype=torch.float32, device=x.device)
    d_x = torch.empty(B, T, dtype=torch.float32, device=x.device) if need_dx else None
    _ffi.check(_ffi.lib().ddsp_hip_fft_convolve_backward(ptr(x), int(x_is_u01), ptr(taps), ptr(grad), ptr(d_x), ptr(d_taps),
                                                         B, F, T // F, N, _ffi.stream_of(x)))
    return d_x, d_taps


def _mag_taps_bwd(d_taps, c, ld, scale, mode, hw):
    B, F, N = d_taps.shape
    n = N // 2 + 1
    d_c = torch.empty(B, F, n, dtype=torch.float32, device=d_taps.device)
    _ffi.check(_ffi.lib().ddsp_hip_impulse_response_backward(ptr(d_taps), ptr(c), ld, _ffi.ACT_EXP, float(scale), int(mode), ptr(hw),
                                                             B * F, n, ptr(ir_table(n, d_taps.device)), ptr(d_c), None,
                                                             _ffi.stream_of(d_taps)))
    return d_c


def _allpass_taps_bwd(d_taps, c, ld):
    B, F, N = d_taps.shape
    n = N // 2 + 1
    dev = d_taps.device
    d_c = torch.empty(B, F, n, dtype=torch.float32, device=dev)

    def call(d_re, d_im):
        return _ffi.lib().ddsp_hip_allpass_taps_backward(ptr(d_taps), ptr(c), ld, B * F, n, ptr(ir_table(n, dev)), ptr(d_c),
                                                         ptr(d_re), ptr(d_im), _ffi.stream_of(d_taps))
    # one launch at 256 bins (the activation's adjoint in the tap adjoint's last stage); the other forms go through (d re, d im)
    # and say so (DDSP_HIP_EWS = -4) when the scratch is missing
    rc = call(None, None) if n == 256 else -4
    if rc == -4:
        d_re = torch.empty(B * F, n, dtype=torch.float32, device=dev)
        rc = call(d_re, torch.empty_like(d_re))
    _ffi.check(rc)
    return d_c


def _sum_cot(T_shape, device, *gs):
    """the cotangent of a branch: the sum of the output cotangents that reach it (None = zero)"""
    gs = [_f32c(g.reshape(T_shape)) for g in gs if g is not None]
    if not gs:
        return None
    out = gs[0]
    for g in gs[1:]:
        out = out + g
    return out


class CombSubTailFunction(torch.autograd.Function):
    """The CombSub DSP tail (vocoder.py:834-862) as ONE autograd node: forward = ``ddsp_hip_combsub_synth`` (the fused inference
    call, all three outputs), backward = the adjoint kernels on the intermediates that call left in its workspace."""

    @staticmethod
    def forward(ctx, f0, state, cg, ch, cn, nz, noise_is_u01, sr, hop, layout):
        B, F = f0.shape
        T = F * hop
        n = cg.shape[-1]
        g_, ldg = _rows(cg.detach(), n)
        h_, ldh = _rows(ch.detach(), n)
        n_, ldn = _rows(cn.detach(), n)
        dev = f0.device
        ws, need = _workspace(B, F, hop, n, dev)
        signal, harm, nzo = _outputs(B, T, dev, True)
        tab = ir_table(n, dev)
        _ffi.check(_ffi.lib().ddsp_hip_combsub_synth(
            ptr(f0), ptr(state.initial_phase), ptr(state.phase0), ptr(g_), ldg, ptr(h_), ldh, ptr(n_), ldn,
            ptr(nz), int(noise_is_u01), B, F, hop, float(sr), int(state.infer), n, n, n, ptr(tab), ptr(tab), ptr(tab),
            ptr(signal), ptr(harm), ptr(nzo), ptr(ws), need, 0, _ffi.stream_of(f0), None, 0, 0))
        N = 2 * (n - 1)
        ctx.save_for_backward(f0, g_, h_, n_, nz, ws)
        ctx.cfg = (ldg, ldh, ldn, bool(noise_is_u01), float(sr), int(hop), layout, (B, F, T, N))
        ctx.set_materialize_grads(False)                        # an output nobody differentiates sends None, not a [B, T] of zeros
        return signal, harm, nzo

    @staticmethod
    def backward(ctx, g_sig, g_harm, g_nz):
        f0, cg, ch, cn, nz, ws = ctx.saved_tensors
        ldg, ldh, ldn, u01, sr, hop, lay, (B, F, T, N) = ctx.cfg
        dev = f0.device
        gh = _sum_cot((B, T), dev, g_sig, g_harm)
        gn = _sum_cot((B, T), dev, g_sig, g_nz)
        if gh is None and gn is None:
            return (None,) * 10
        n = N // 2 + 1
        lib = _ffi.lib()
        # three launches (csrc/api.hip, ddsp_hip_combsub_tail_backward): harmonic filter's adjoint | all-pass + noise filter tap
        # gradients (two jobs) | the three tap-synthesis adjoints (three jobs; the dynamic window's half widths from f0 in the kernel,
        # the all-pass activation's adjoint in that job's last stage)
        need = lib.ddsp_hip_combsub_tail_backward_ws_bytes(B, F, hop, n)
        bws = torch.empty(need, dtype=torch.uint8, device=dev)
        d_cg = torch.empty(B, F, n, dtype=torch.float32, device=dev) if gh is not None else None
        d_ch = torch.empty(B, F, n, dtype=torch.float32, device=dev) if gh is not None else None
        d_cn = torch.empty(B, F, n, dtype=torch.float32, device=dev) if gn is not None else None
        _ffi.check(lib.ddsp_hip_combsub_tail_backward(
            ptr(f0), ptr(cg), ldg, ptr(ch), ldh, ptr(cn), ldn, ptr(nz), int(u01), ptr(ws), ptr(gh), ptr(gn), B, F, hop, sr, n,
            ptr(ir_table(n, dev)), ptr(d_cg), ptr(d_ch), ptr(d_cn), ptr(bws), need, _ffi.stream_of(f0)))
        return None, None, d_cg, d_ch, d_cn, None, None, None, None, None


class SinsTailFunction(torch.autograd.Function):
    """The Sins DSP tail (vocoder.py:580-611) as one autograd node, as CombSubTailFunction."""

    @staticmethod
    def forward(ctx, f0, state, ca, cg, cn, nz, noise_is_u01, sr, hop, layout):
        B, F = f0.shape
        T = F * hop
        H, n = ca.shape[-1], cg.shape[-1]
        a_, lda = _rows(ca.detach(), H)
        g_, ldg = _rows(cg.detach(), n)
        n_, ldn = _rows(cn.detach(), n)
        dev = f0.device
        ws, need = _workspace(B, F, hop, n, dev)
        signal, harm, nzo = _outputs(B, T, dev, True)
        tab = ir_table(n, dev)
        _ffi.check(_ffi.lib().ddsp_hip_sins_synth(
            ptr(f0), ptr(state.initial_phase), ptr(state.phase0), ptr(a_), lda, ptr(g_), ldg, ptr(n_), ldn,
            ptr(nz), int(noise_is_u01), B, F, hop, float(sr), int(state.infer), H, n, n, ptr(tab), ptr(tab),
            ptr(signal), ptr(harm), ptr(nzo), ptr(ws), need, 0, _ffi.stream_of(f0), None, 0, 0))
        ctx.save_for_backward(f0, a_, g_, n_, nz, ws)
        ctx.cfg = (lda, ldg, ldn, bool(noise_is_u01), float(sr), int(hop), layout, state, (B, F, T, 2 * (n - 1), H))
        ctx.set_materialize_grads(False)
        return signal, harm, nzo

    @staticmethod
    def backward(ctx, g_sig, g_harm, g_nz):
        f0, ca, cg, cn, nz, ws = ctx.saved_tensors
        lda, ldg, ldn, u01, sr, hop, lay, state, (B, F, T, N, H) = ctx.cfg
        sinus = _ws_view(ws, lay[0], (B, T))
        # (the noise taps: only their shape is used below -- no input gradient is asked of that filter, so its adjoint reads no
        # taps; the workspace may hold them as half rows, ddsp_hip.h)
        taps_ap, taps_nz = _ws_view(ws, lay[2], (B, F, N)), _ws_view(ws, lay[4], (B, F, N))
        gh = _sum_cot((B, T), f0.device, g_sig, g_harm)
        gn = _sum_cot((B, T), f0.device, g_sig, g_nz)
        d_ca = d_cg = d_cn = None
        lib = _ffi.lib()
        if gh is not None:
            d_sin, d_taps_ap = _fir_bwd(sinus, False, taps_ap, gh, True)                     # vocoder.py:597-600 backwards
            d_cg = _allpass_taps_bwd(d_taps_ap, cg, ldg)
            scratch = torch.empty(lib.ddsp_hip_sinusoid_bank_backward_scratch_bytes(B, F, H), dtype=torch.uint8, device=f0.device)
            d_ca = torch.empty(B, F, H, dtype=torch.float32, device=f0.device)
            _ffi.check(lib.ddsp_hip_sinusoid_bank_backward(ptr(f0), ptr(state.initial_phase), ptr(state.phase0), ptr(ca), lda,
                                                           ptr(d_sin), B, F, hop, H, sr, int(state.infer), ptr(scratch), ptr(d_ca),
                                                           _ffi.stream_of(f0)))                # :585-594
        if gn is not None:
            _, d_taps_nz = _fir_bwd(nz, u01, taps_nz, gn, False)                             # :603-607
            d_cn = _mag_taps_bwd(d_taps_nz, cn, ldn, 1.0 / 128.0, _ffi.MODE_HANN, None)
        return None, None, d_ca, d_cg, d_cn, None, None, None, None, None


def _fused_train_ok(f0_frames, hop, *ctrls):
    """float32 controls on the device, hop 512 -- and what the library's own predicate says (ddsp_hip_tail_layout)"""
    return int(hop) == 512 and all(c.dtype == torch.float32 and c.dim() == 3 for c in ctrls)


def _combsub_synth_train(f0_frames, state, group_delay, harmonic_magnitude, noise_magnitude, noise, sampling_rate,
                         block_size, noise_is_u01):
    """CombSub DSP tail as a composition of differentiable primitives (training, solver.py:93-103): the same kernels as
    the fused entry point, intermediates kept for the backward pass.  Returns (signal, harmonic, noise)."""
    from .core import fft_convolve, fft_convolve_add
    f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
    B, F = f0.shape
    nz = _f32c(noise.reshape(B, -1))
    if noise_is_u01:
        nz = nz * 2 - 1                                                                        # :854
    # the noise branch does not meet the harmonic chain before the final sum: second stream (forward and backward)
    noise_f, join = _ffi.on_aux_stream(lambda: fft_convolve(nz, MagnitudeTapsFunction.apply(
        noise_magnitude, 1.0 / 128.0, _ffi.MODE_HANN, None)), f0, B * F, defer_join=True)                     # :855-858
    if _EARLY_JOIN:
        join()
    try:
        comb = combtooth(f0_frames, state, sampling_rate, block_size)                         # vocoder.py:839-840
        h1 = fft_convolve(comb, AllpassTapsFunction.apply(group_delay))                       # :843-846
        hw = (1.5 * float(sampling_rate)) / (f0 + 1e-3)                                       # :851
        taps_h = MagnitudeTapsFunction.apply(harmonic_magnitude, 1.0, _ffi.MODE_DYNAMIC, hw)  # :847-851
    finally:
        join()                        # the noise branch meets the chain here -- also when a launch above raised: the branch still
                                      # reads the noise and the controls, and the caller's stream must stay ordered behind it
    signal, harmonic = fft_convolve_add(h1, taps_h, noise_f)                                  # :860: the sum rides in the filter
    return signal, harmonic, noise_f


def _sins_synth_train(f0_frames, state, amplitudes, group_delay, noise_magnitude, noise, sampling_rate, block_size,
                      noise_is_u01):
    """Sins DSP tail as a composition of differentiable primitives (training).  Returns (signal, harmonic, noise)."""
    from .core import fft_convolve, fft_convolve_add
    B = f0_frames.shape[0]
    nz = _f32c(noise.reshape(B, -1))
    if noise_is_u01:
        nz = nz * 2 - 1                                                                                # :603
    noise_f, join = _ffi.on_aux_stream(lambda: fft_convolve(nz, MagnitudeTapsFunction.apply(
        noise_magnitude, 1.0 / 128.0, _ffi.MODE_HANN, None)), nz, B * group_delay.shape[1], defer_join=True)       # :604-607
    if _EARLY_JOIN:
        join()
    try:
        sinus = SinusoidBankFunction.apply(f0_frames, state, amplitudes, sampling_rate, block_size)   # vocoder.py:585-594
        taps_ap = AllpassTapsFunction.apply(group_delay)
    finally:
        join()                                                   # always: see _combsub_synth_train
    signal, harmonic = fft_convolve_add(sinus, taps_ap, noise_f)                                      # :597-600, :609
    return signal, harmonic, noise_f


def combsub_synth(f0_frames, state: PhaseState, group_delay, harmonic_magnitude, noise_magnitude, noise,
                  sampling_rate, block_size, noise_is_u01=False, want_components=True, fir_impl=_ffi.FIR_AUTO,
                  noise_seed=None, noise_offset=0, signal_out=None):
    """DSP tail of ``CombSub.forward`` (vocoder.py:834-862) from raw controls; ``noise=None``: the uniform draw happens
    inside the noise filter from ``(noise_seed, noise_offset)`` (see ``sins_synth``).  With gradients enabled and a control
    that requires grad the differentiable composition is used (the filters at every hop and bin count; their fast adjoint kernels at hop 512, n_mag <= 257)."""
    _ffi.check_device(f0_frames, group_delay, harmonic_magnitude, noise_magnitude, noise, state.phase0)
    _need_seed(noise, noise_seed)
    if noise is None and torch.is_grad_enabled() and any(c.requires_grad for c in (group_delay, harmonic_magnitude, noise_magnitude)):
        noise, noise_is_u01 = uniform_noise(f0_frames.shape[0], f0_frames.shape[1] * int(block_size), noise_seed, noise_offset,
                                            f0_frames.device), True
    if torch.is_grad_enabled() and any(c.requires_grad for c in (group_delay, harmonic_magnitude, noise_magnitude)):
        f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
        B, F = f0.shape
        n = group_delay.shape[-1]
        lay = None
        if _fused_train_ok(f0_frames, block_size, group_delay, harmonic_magnitude, noise_magnitude) and fir_impl == _ffi.FIR_AUTO \
                and harmonic_magnitude.shape[-1] == n and noise_magnitude.shape[-1] == n and not _TRAIN_COMPOSED:
            lay = _tail_layout(True, B, F, int(block_size), n, n, n)
        if lay is not None:
            return CombSubTailFunction.apply(f0, state, group_delay, harmonic_magnitude, noise_magnitude,
                                             _f32c(noise.reshape(B, -1)), noise_is_u01, sampling_rate, int(block_size), lay)
        return _combsub_synth_train(f0_frames, state, group_delay, harmonic_magnitude, noise_magnitude, noise,
                                    sampling_rate, block_size, noise_is_u01)
    f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
    B, F = f0.shape
    hop = int(block_size)
    T = F * hop
    n_ap, n_h, n_nz = group_delay.shape[-1], harmonic_magnitude.shape[-1], noise_magnitude.shape[-1]
    cg, ldg = _rows(group_delay, n_ap)
    ch, ldh = _rows(harmonic_magnitude, n_h)
    cn, ldn = _rows(noise_magnitude, n_nz)
    dev = f0.device
    nz, is_u01, seed, offset = _noise_arg(noise, noise_seed, noise_offset, noise_is_u01, B, T, hop, n_nz, fir_impl, dev)
    ws, need = _workspace(B, F, hop, max(n_ap, n_h, n_nz), dev)
    signal, harm, nzo = _outputs(B, T, dev, want_components)
    if signal_out is not None:
        # (inference only) the waveform goes where the caller wants it -- e.g. this rank's slice of a gather's result tensor
        # (sharding.gather_utterances(out=)): no copy between the synthesis and the collective
        if tuple(signal_out.shape) != (B, T) or signal_out.dtype != torch.float32 or signal_out.device != dev or not signal_out.is_contiguous():
            raise ValueError("signal_out must be a contiguous float32 [%d, %d] tensor on %s" % (B, T, dev))
        signal = signal_out
    _ffi.check(_ffi.lib().ddsp_hip_combsub_synth(
        ptr(f0), ptr(state.initial_phase), ptr(state.phase0), ptr(cg), ldg, ptr(ch), ldh, ptr(cn), ldn,
        ptr(nz), int(is_u01), B, F, hop, float(sampling_rate), int(state.infer), n_ap, n_h, n_nz,
        ptr(ir_table(n_ap, dev)), ptr(ir_table(n_h, dev)), ptr(ir_table(n_nz, dev)),
        ptr(signal), ptr(harm), ptr(nzo), ptr(ws), need, int(fir_impl), _ffi.stream_of(f0), _ffi.aux_stream_of(f0, B * F),
        seed, offset))
    return signal, harm, nzo


class StreamingCombSub:
    """The CombSub tail for a real-time caller (gui.py:118-133: the same small shape every audio callback): ``phase`` and
    ``synth`` of one FIXED shape with every buffer -- phase state, workspace, outputs -- allocated once and every pointer
    bound once.  A call is then two C calls and nothing else on the host: at B = 1, a second of audio, the functional API
    above spends ~25 us of Python (six ``torch.empty``, argument marshalling) around ~35 us of GPU work.

    The numbers are those of ``phase`` + ``combsub_synth`` bit for bit (same library calls).  The returned tensors are the
    session's own buffers: the next call overwrites them -- copy what must outlive it (the GUI copies to the host anyway).
    One session per host thread / stream; controls must be float32 with a contiguous last dimension."""

    def __init__(self, B, F, n_mag_allpass, n_mag_harmonic, n_mag_noise, sampling_rate, block_size, device, infer=True,
                 want_components=False):
        self.B, self.F, self.hop, self.sr, self.infer = int(B), int(F), int(block_size), float(sampling_rate), bool(infer)
        self.n = (int(n_mag_allpass), int(n_mag_harmonic), int(n_mag_noise))
        dev = torch.device(device)
        T = self.F * self.hop
        self._sums = torch.empty(B, F, dtype=torch.float64, device=dev)
        self.state = PhaseState(torch.empty(B, F, dtype=torch.float64, device=dev), torch.empty(B, F, 1, dtype=torch.float32, device=dev),
                                None, self.infer)
        self._ws, self._need = _workspace(B, F, self.hop, max(self.n), dev)
        self.signal, self.harmonic, self.noise = _outputs(B, T, dev, want_components)
        self._tables = tuple(ir_table(n, dev) for n in self.n)
        self._lib = _ffi.lib()
        _ffi.check_device(self.signal)
        self._dev = self.signal.device
        # the buffers are ordered on the stream the session was made on: a call from another stream would race with them
        self._stream = _ffi.stream_of(self.signal)

    def _check(self, t, shape, what):
        """one cheap look per argument and call: a wrong shape / dtype / device / stride handed to the C ABI as a raw pointer is a
        silent out-of-bounds read on the GPU, not an exception"""
        if t.device != self._dev or t.dtype != torch.float32 or tuple(t.shape) != shape or t.stride(-1) != 1:
            raise ValueError("StreamingCombSub: %s must be a float32 %s tensor on %s with a contiguous last dimension (got %s %s on %s, "
                             "strides %s)" % (what, shape, self._dev, t.dtype, tuple(t.shape), t.device, tuple(t.stride())))

    def _check_ctrl(self, t, n, what):
        self._check(t, (self.B, self.F, n), what)
        if t.stride(1) < n or (self.B > 1 and t.stride(0) != self.F * t.stride(1)):
            raise ValueError("StreamingCombSub: %s needs one uniform frame stride >= %d (a torch.split view is fine)" % (what, n))

    def _check_stream(self, t):
        if _ffi.stream_of(t) != self._stream:
            raise RuntimeError("StreamingCombSub: called on a stream other than the one the session was created on")

    def phase(self, f0_frames):
        """``synth.phase`` into the session's state (what ``Unit2Control`` needs is ``.phase_frames``)"""
        if f0_frames.numel() != self.B * self.F or not f0_frames.is_contiguous():
            raise ValueError("StreamingCombSub: f0_frames must be a contiguous [%d, %d(, 1)] tensor" % (self.B, self.F))
        f0 = f0_frames.reshape(self.B, self.F)
        self._check(f0, (self.B, self.F), "f0_frames")
        self._check_stream(f0)
        st = self.state
        _ffi.check(self._lib.ddsp_hip_phase(f0.data_ptr(), None, self.B, self.F, self.hop, self.sr, int(self.infer),
                                            self._sums.data_ptr(), st.phase0.data_ptr(), st.phase_frames.data_ptr(), None,
                                            _ffi.stream_of(f0)))
        return st

    def synth(self, f0_frames, group_delay, harmonic_magnitude, noise_magnitude, noise, noise_is_u01=False):
        """``combsub_synth`` on the session's state and buffers -> ``signal`` (and the components, if asked for)"""
        if f0_frames.numel() != self.B * self.F or not f0_frames.is_contiguous():
            raise ValueError("StreamingCombSub: f0_frames must be a contiguous [%d, %d(, 1)] tensor" % (self.B, self.F))
        f0 = f0_frames.reshape(self.B, self.F)
        self._check(f0, (self.B, self.F), "f0_frames")
        self._check_ctrl(group_delay, self.n[0], "group_delay")
        self._check_ctrl(harmonic_magnitude, self.n[1], "harmonic_magnitude")
        self._check_ctrl(noise_magnitude, self.n[2], "noise_magnitude")
        self._check(noise.reshape(self.B, -1) if noise.numel() == self.B * self.F * self.hop and noise.is_contiguous() else noise,
                    (self.B, self.F * self.hop), "noise")
        self._check_stream(f0)
        t = self._tables
        _ffi.check(self._lib.ddsp_hip_combsub_synth(
            f0.data_ptr(), None, self.state.phase0.data_ptr(), group_delay.data_ptr(), group_delay.stride(1),
            harmonic_magnitude.data_ptr(), harmonic_magnitude.stride(1), noise_magnitude.data_ptr(), noise_magnitude.stride(1),
            noise.data_ptr(), int(noise_is_u01), self.B, self.F, self.hop, self.sr, int(self.infer), self.n[0], self.n[1], self.n[2],
            t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), self.signal.data_ptr(), ptr(self.harmonic), ptr(self.noise),
            self._ws.data_ptr(), self._need, 0, _ffi.stream_of(f0), None, 0, 0))
        return self.signal if self.harmonic is None else (self.signal, self.harmonic, self.noise)


# ---- CombSubFast / CombSubSuperFast (vocoder.py:613-786) ------------------------------------------
@dataclass
class FastSourceState:
    """Output of ``fast_source``: ``phase_frames [B,F,1]`` for Unit2Control and ``rad_acc [B,F]``, the
    float32 frame-rate phase accumulator the exciter restarts from (vocoder.py:646)."""
    rad_acc: torch.Tensor
    phase_frames: torch.Tensor
    combtooth: Optional[torch.Tensor] = None


def fast_source(f0_frames, sampling_rate, block_size, want_combtooth=False) -> FastSourceState:
    """``CombSubSuperFast.fast_source_gen`` (vocoder.py:639-651)."""
    _ffi.check_device(f0_frames)
    f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
    B, F = f0.shape
    hop = int(block_size)
    dev = f0.device
    rad_acc = torch.empty(B, F, dtype=torch.float32, device=dev)
    pf = torch.empty(B, F, 1, dtype=torch.float32, device=dev)
    comb = torch.empty(B, F * hop, dtype=torch.float32, device=dev) if want_combtooth else None
    _ffi.check(_ffi.lib().ddsp_hip_fast_source(ptr(f0), B, F, hop, float(sampling_rate), ptr(rad_acc), ptr(pf),
                                               ptr(comb), _ffi.stream_of(f0)))
    return FastSourceState(rad_acc, pf, comb)


def _stft_ws(B, F, hop, device):
    need = _ffi.lib().ddsp_hip_stft_workspace_bytes(B, F, hop)
    return torch.empty(need, dtype=torch.uint8, device=device), need


def stft_filter(exciter, noise, harmonic_magnitude, harmonic_phase, noise_magnitude, noise_phase, window,
                block_size, noise_scale=1.0 / 128.0, pad_reflect=True, normalize=True, noise_is_u01=False):
    """The shared spectral-filtering tail (vocoder.py:661-708 / :758-784) on explicit exciter and noise
    signals ``[B,T]``; ``noise_phase`` may be None (zero-phase noise filter)."""
    _ffi.check_device(exciter, noise, harmonic_magnitude, harmonic_phase, noise_magnitude, window)
    B, T = exciter.shape
    hop = int(block_size)
    F = T // hop
    win = window.numel()
    n = win // 2 + 1
    hm, ldhm = _rows(harmonic_magnitude, n)
    hp, ldhp = _rows(harmonic_phase, n)
    nm, ldnm = _rows(noise_magnitude, n)
    npz, ldnp = (None, 0) if noise_phase is None else _rows(noise_phase, n)
    out = torch.empty(B, T, dtype=torch.float32, device=exciter.device)
    _ffi.check(_ffi.lib().ddsp_hip_stft_filter(
        ptr(_f32c(exciter)), ptr(_f32c(noise)), int(noise_is_u01), ptr(hm), ldhm, ptr(hp), ldhp, ptr(nm), ldnm,
        ptr(npz), ldnp, float(noise_scale), ptr(_f32c(window)), win, int(pad_reflect), int(normalize), B, F, hop,
        ptr(out), _ffi.stream_of(exciter)))
    return out


def stft_filter_backward(grad_signal, exciter, noise, harmonic_magnitude, harmonic_phase, noise_magnitude, noise_phase,
                         window, block_size, noise_scale=1.0 / 128.0, pad_reflect=True, normalize=True,
                         noise_is_u01=False):
    """Gradients of ``stft_filter`` w.r.t. its control streams for the cotangent ``grad_signal [B,T]``:
    ``(d_hmag, d_hphase, d_nmag, d_nphase|None)``, each ``[B,F,n]``."""
    _ffi.check_device(grad_signal, exciter, noise, harmonic_magnitude, harmonic_phase, noise_magnitude, window)
    B, T = exciter.shape
    hop = int(block_size)
    F = T // hop
    win = window.numel()
    n = win // 2 + 1
    hm, ldhm = _rows(harmonic_magnitude, n)
    hp, ldhp = _rows(harmonic_phase, n)
    nm, ldnm = _rows(noise_magnitude, n)
    npz, ldnp = (None, 0) if noise_phase is None else _rows(noise_phase, n)
    dev = exciter.device
    d_hm, d_hp, d_nm = (torch.empty(B, F, n, dtype=torch.float32, device=dev) for _ in range(3))
    d_np = None if noise_phase is None else torch.empty(B, F, n, dtype=torch.float32, device=dev)
    _ffi.check(_ffi.lib().ddsp_hip_stft_filter_backward(
        ptr(_f32c(exciter)), ptr(_f32c(noise)), int(noise_is_u01), ptr(hm), ldhm, ptr(hp), ldhp, ptr(nm), ldnm,
        ptr(npz), ldnp, float(noise_scale), ptr(_f32c(window)), win, int(pad_reflect), int(normalize),
        ptr(_f32c(grad_signal.reshape(B, T))), B, F, hop, ptr(d_hm), ptr(d_hp), ptr(d_nm), ptr(d_np),
        _ffi.stream_of(exciter)))
    return d_hm, d_hp, d_nm, d_np


class StftFilterFunction(torch.autograd.Function):
    """``stft_filter`` with autograd w.r.t. the control streams (exciter, noise and window are data).  Training
    back-propagates through CombSubFast / CombSubSuperFast.forward into Unit2Control (solver.py:93-103)."""

    @staticmethod
    def forward(ctx, exciter, noise, hmag, hphase, nmag, nphase, window, block_size, noise_scale, pad_reflect,
                normalize, noise_is_u01):
        ctx.save_for_backward(exciter, noise, hmag, hphase, nmag, nphase if nphase is not None else hmag.new_empty(0),
                              window)
        ctx.cfg = (int(block_size), float(noise_scale), bool(pad_reflect), bool(normalize), bool(noise_is_u01),
                   nphase is not None)
        return stft_filter(exciter, noise, hmag.detach(), hphase.detach(), nmag.detach(),
                           None if nphase is None else nphase.detach(), window, block_size, noise_scale=noise_scale,
                           pad_reflect=pad_reflect, normalize=normalize, noise_is_u01=noise_is_u01)

    @staticmethod
    def backward(ctx, grad_signal):
        exciter, noise, hmag, hphase, nmag, nphase, window = ctx.saved_tensors
        hop, scale, reflect, normalize, u01, has_np = ctx.cfg
        d_hm, d_hp, d_nm, d_np = stft_filter_backward(grad_signal.contiguous(), exciter, noise, hmag, hphase, nmag,
                                                      nphase if has_np else None, window, hop, noise_scale=scale,
                                                      pad_reflect=reflect, normalize=normalize, noise_is_u01=u01)
        return None, None, d_hm, d_hp, d_nm, d_np, None, None, None, None, None, None


def combsubfast_synth(f0_frames, state: PhaseState, harmonic_magnitude, harmonic_phase, noise_magnitude, noise,
                      window, sampling_rate, block_size, noise_is_u01=False):
    """DSP tail of ``CombSubFast.forward`` (vocoder.py:758-784) from raw controls -> ``signal [B,T]``."""
    _ffi.check_device(f0_frames, harmonic_magnitude, harmonic_phase, noise_magnitude, noise, window, state.phase0)
    f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
    B, F = f0.shape
    hop = int(block_size)
    T = F * hop
    n = hop + 1
    if window.numel() != 2 * hop:
        raise ValueError("CombSubFast window must hold 2 * block_size samples")
    nz = _f32c(noise.reshape(B, T))
    if torch.is_grad_enabled() and any(c.requires_grad for c in (harmonic_magnitude, harmonic_phase, noise_magnitude)):
        # training: exciter materialised once (it carries no gradient), the spectral tail through autograd
        comb = combtooth(f0_frames, state, sampling_rate, block_size)
        return StftFilterFunction.apply(comb, nz, harmonic_magnitude, harmonic_phase, noise_magnitude, None, window,
                                        hop, 1.0 / 128.0, False, False, noise_is_u01)
    hm, ldhm = _rows(harmonic_magnitude, n)
    hp, ldhp = _rows(harmonic_phase, n)
    nm, ldnm = _rows(noise_magnitude, n)
    dev = f0.device
    ws, need = _stft_ws(B, F, hop, dev)
    signal = torch.empty(B, T, dtype=torch.float32, device=dev)
    _ffi.check(_ffi.lib().ddsp_hip_combsubfast_synth(
        ptr(f0), ptr(state.initial_phase), ptr(state.phase0), ptr(hm), ldhm, ptr(hp), ldhp, ptr(nm), ldnm,
        ptr(nz), int(noise_is_u01), ptr(_f32c(window)), B, F, hop, float(sampling_rate), int(state.infer),
        ptr(signal), ptr(ws), need, _ffi.stream_of(f0)))
    return signal


def combsubsuperfast_synth(f0_frames, state: FastSourceState, harmonic_magnitude, harmonic_phase, noise_magnitude,
                           noise_phase, noise, window, sampling_rate, block_size):
    """DSP tail of ``CombSubSuperFast.forward`` (vocoder.py:661-708) from raw controls; ``noise [B,T]`` is the
    standard-normal draw (:687) -> ``signal [B,T]``."""
    _ffi.check_device(f0_frames, harmonic_magnitude, harmonic_phase, noise_magnitude, noise_phase, noise, window,
                      state.rad_acc)
    f0 = _f32c(f0_frames.reshape(f0_frames.shape[0], -1))
    B, F = f0.shape
    hop = int(block_size)
    T = F * hop
    win = window.numel()
    n = win // 2 + 1
    nz = _f32c(noise.reshape(B, T))
    if torch.is_grad_enabled() and any(c.requires_grad for c in (harmonic_magnitude, harmonic_phase, noise_magnitude,
                                                                  noise_phase)):
        comb = fast_source(f0_frames, sampling_rate, block_size, want_combtooth=True).combtooth
        return StftFilterFunction.apply(comb, nz, harmonic_magnitude, harmonic_phase, noise_magnitude, noise_phase,
                                        window, hop, 1.0 / 128.0, T > win // 2, True, False)
    hm, ldhm = _rows(harmonic_magnitude, n)
    hp, ldhp = _rows(harmonic_phase, n)
    nm, ldnm = _rows(noise_magnitude, n)
    npz, ldnp = _rows(noise_phase, n)
    dev = f0.device
    ws, need = _stft_ws(B, F, hop, dev)
    signal = torch.empty(B, T, dtype=torch.float32, device=dev)
    _ffi.check(_ffi.lib().ddsp_hip_combsubsuperfast_synth(
        ptr(f0), ptr(state.rad_acc), ptr(hm), ldhm, ptr(hp), ldhp, ptr(nm), ldnm, ptr(npz), ldnp, ptr(nz),
        ptr(_f32c(window)), win, B, F, hop, float(sampling_rate), ptr(signal), ptr(ws), need, _ffi.stream_of(f0)))
    return signal


class StreamingCombSubSuperFast:
    """The CombSubSuperFast tail for the real-time caller (gui.py:118-133 runs THIS model, configs/combsub.yaml:19): ``source``
    and ``synth`` of one fixed shape with every buffer allocated and every pointer bound once, as ``StreamingCombSub`` -- a call
    is two C calls (``ddsp_hip_fast_source`` before ``Unit2Control``, ``ddsp_hip_combsubsuperfast_synth`` behind it) and nothing
    else on the host.  The numbers are those of ``fast_source`` + ``combsubsuperfast_synth`` bit for bit.  The returned tensors are
    the session's own buffers (the next call overwrites them); one session per host thread / stream; float32 controls
    ``[B, F, win // 2 + 1]`` with a contiguous last dimension and one uniform frame stride (``torch.split`` views are fine)."""

    def __init__(self, B, F, window, sampling_rate, block_size, device):
        self.B, self.F, self.hop, self.sr = int(B), int(F), int(block_size), float(sampling_rate)
        dev = torch.device(device)
        self.window = _f32c(window.to(dev))
        self.win = self.window.numel()
        self.n = self.win // 2 + 1
        self.state = FastSourceState(torch.empty(B, F, dtype=torch.float32, device=dev),
                                     torch.empty(B, F, 1, dtype=torch.float32, device=dev), None)
        self._ws, self._need = _stft_ws(B, F, self.hop, dev)
        self.signal = torch.empty(B, F * self.hop, dtype=torch.float32, device=dev)
        self._lib = _ffi.lib()
        _ffi.check_device(self.signal)
        self._dev = self.signal.device
        self._stream = _ffi.stream_of(self.signal)

    def _f0(self, f0_frames):
        if f0_frames.numel() != self.B * self.F or not f0_frames.is_contiguous() or f0_frames.dtype != torch.float32 \
                or f0_frames.device != self._dev:
            raise ValueError("StreamingCombSubSuperFast: f0_frames must be a contiguous float32 [%d, %d(, 1)] tensor on %s"
                             % (self.B, self.F, self._dev))
        if _ffi.stream_of(f0_frames) != self._stream:
            raise RuntimeError("StreamingCombSubSuperFast: called on a stream other than the one the session was created on")
        return f0_frames

    def _ctrl(self, t, what):
        if t.device != self._dev or t.dtype != torch.float32 or tuple(t.shape) != (self.B, self.F, self.n) or t.stride(2) != 1 \
                or t.stride(1) < self.n or (self.B > 1 and t.stride(0) != self.F * t.stride(1)):
            raise ValueError("StreamingCombSubSuperFast: %s must be a float32 [%d, %d, %d] tensor on %s with a contiguous last dimension "
                             "and one uniform frame stride (got %s %s, strides %s)"
                             % (what, self.B, self.F, self.n, self._dev, t.dtype, tuple(t.shape), tuple(t.stride())))
        return t

    def source(self, f0_frames):
        """``fast_source`` into the session's state (what ``Unit2Control`` needs is ``.phase_frames``)"""
        f0 = self._f0(f0_frames)
        st = self.state
        _ffi.check(self._lib.ddsp_hip_fast_source(f0.data_ptr(), self.B, self.F, self.hop, self.sr, st.rad_acc.data_ptr(),
                                                  st.phase_frames.data_ptr(), None, self._stream))
        return st

    def synth(self, f0_frames, harmonic_magnitude, harmonic_phase, noise_magnitude, noise_phase, noise):
        """``combsubsuperfast_synth`` on the session's state and buffers -> ``signal [B, T]``"""
        f0 = self._f0(f0_frames)
        hm, hp = self._ctrl(harmonic_magnitude, "harmonic_magnitude"), self._ctrl(harmonic_phase, "harmonic_phase")
        nm, npz = self._ctrl(noise_magnitude, "noise_magnitude"), self._ctrl(noise_phase, "noise_phase")
        if noise.device != self._dev or noise.dtype != torch.float32 or noise.numel() != self.B * self.F * self.hop \
                or not noise.is_contiguous():
            raise ValueError("StreamingCombSubSuperFast: noise must be a contiguous float32 [%d, %d] tensor on %s"
                             % (self.B, self.F * self.hop, self._dev))
        _ffi.check(self._lib.ddsp_hip_combsubsuperfast_synth(
            f0.data_ptr(), self.state.rad_acc.data_ptr(), hm.data_ptr(), hm.stride(1), hp.data_ptr(), hp.stride(1), nm.data_ptr(),
            nm.stride(1), npz.data_ptr(), npz.stride(1), noise.data_ptr(), self.window.data_ptr(), self.win, self.B, self.F, self.hop,
            self.sr, self.signal.data_ptr(), self._ws.data_ptr(), self._need, self._stream))
        return self.signal
