"""ctypes binding of libddsp_hip.so (C ABI declared in include/ddsp_hip.h).

The library is the only compute path: there is no CPU fallback.  ``lib()`` raises if the shared
object cannot be loaded (or built with hipcc), and ``check_device`` rejects host tensors.
"""
import ctypes
import os
import threading

import torch

from . import build as _build

c_int, c_long, c_float, c_double, c_size_t = (ctypes.c_int, ctypes.c_long, ctypes.c_float,
                                              ctypes.c_double, ctypes.c_size_t)
P = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/ddsp_hip.h one to one
SIGNATURES = {
    "ddsp_hip_version": (c_int, []),
    "ddsp_hip_error_string": (ctypes.c_char_p, [c_int]),
    "ddsp_hip_set_tuning": (c_int, [ctypes.c_char_p, c_long]),
    "ddsp_hip_get_tuning": (c_long, [ctypes.c_char_p]),
    "ddsp_hip_window_impulse_response": (c_int, [P, c_int, P, c_long, c_int, P, P]),
    "ddsp_hip_upsample": (c_int, [P, c_int, c_int, c_int, c_int, P, P]),
    "ddsp_hip_remove_above_fmax": (c_int, [P, P, c_long, c_int, c_float, c_int, P, P]),
    "ddsp_hip_phase": (c_int, [P, P, c_int, c_int, c_int, c_double, c_int, P, P, P, P, P]),
    "ddsp_hip_ir_table_bytes": (c_size_t, [c_int]),
    "ddsp_hip_ir_table": (c_int, [c_int, P, P]),
    "ddsp_hip_allpass_response": (c_int, [P, c_long, c_long, c_int, P, P, P]),
    "ddsp_hip_impulse_response": (c_int, [P, c_long, P, c_long, c_int, c_float, c_int, P, c_long, c_int, P, P, P]),
    "ddsp_hip_allpass_taps_scratch_bytes": (c_size_t, [c_long, c_int]),
    "ddsp_hip_allpass_taps": (c_int, [P, c_long, c_long, c_int, P, P, P, c_size_t, P]),
    "ddsp_hip_impulse_response_backward": (c_int, [P, P, c_long, c_int, c_float, c_int, P, c_long, c_int, P, P, P, P]),
    "ddsp_hip_allpass_backward": (c_int, [P, c_long, c_long, c_int, P, P, P, P]),
    "ddsp_hip_allpass_taps_backward": (c_int, [P, P, c_long, c_long, c_int, P, P, P, P, P]),
    "ddsp_hip_fft_convolve": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "ddsp_hip_frequency_filter_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ddsp_hip_frequency_filter": (c_int, [P, P, c_long, P, c_long, c_int, P, c_int, c_int, c_int, c_int, P, P, P,
                                          c_size_t, P]),
    "ddsp_hip_fft_convolve_backward": (c_int, [P, c_int, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "ddsp_hip_sins_synth": (c_int, [P, P, P, P, c_long, P, c_long, P, c_long, P, c_int,
                                    c_int, c_int, c_int, c_double, c_int, c_int, c_int, c_int,
                                    P, P, P, P, P, P, c_size_t, c_int, P, P, ctypes.c_ulonglong, ctypes.c_ulonglong]),
    "ddsp_hip_combsub_synth": (c_int, [P, P, P, P, c_long, P, c_long, P, c_long, P, c_int,
                                       c_int, c_int, c_int, c_double, c_int, c_int, c_int, c_int,
                                       P, P, P, P, P, P, P, c_size_t, c_int, P, P, ctypes.c_ulonglong, ctypes.c_ulonglong]),
    "ddsp_hip_uniform_noise": (c_int, [ctypes.c_ulonglong, ctypes.c_ulonglong, c_int, c_long, P, P]),
    "ddsp_hip_synth_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ddsp_hip_tail_layout": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "ddsp_hip_combsub_tail_backward_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ddsp_hip_combsub_tail_backward": (c_int, [P, P, c_long, P, c_long, P, c_long, P, c_int, P, P, P, c_int, c_int, c_int, c_double,
                                               c_int, P, P, P, P, P, c_size_t, P]),
    "ddsp_hip_combtooth": (c_int, [P, P, P, c_int, c_int, c_int, c_double, c_int, P, P]),
    "ddsp_hip_sinusoid_bank": (c_int, [P, P, P, P, c_long, c_int, c_int, c_int, c_int, c_double, c_int, P, P]),
    "ddsp_hip_sinusoid_bank_backward_scratch_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ddsp_hip_sinusoid_bank_backward": (c_int, [P, P, P, P, c_long, P, c_int, c_int, c_int, c_int, c_double, c_int,
                                                P, P, P]),
    "ddsp_hip_fast_source": (c_int, [P, c_int, c_int, c_int, c_double, P, P, P, P]),
    "ddsp_hip_stft_filter": (c_int, [P, P, c_int, P, c_long, P, c_long, P, c_long, P, c_long, c_float, P, c_int,
                                     c_int, c_int, c_int, c_int, c_int, P, P]),
    "ddsp_hip_stft_filter_backward": (c_int, [P, P, c_int, P, c_long, P, c_long, P, c_long, P, c_long, c_float, P,
                                              c_int, c_int, c_int, P, c_int, c_int, c_int, P, P, P, P, P]),
    "ddsp_hip_combsubfast_synth": (c_int, [P, P, P, P, c_long, P, c_long, P, c_long, P, c_int, P,
                                           c_int, c_int, c_int, c_double, c_int, P, P, c_size_t, P]),
    "ddsp_hip_combsubsuperfast_synth": (c_int, [P, P, P, c_long, P, c_long, P, c_long, P, c_long, P, P, c_int,
                                                c_int, c_int, c_int, c_double, P, P, c_size_t, P]),
    "ddsp_hip_stft_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ddsp_hip_sine_source": (c_int, [P, c_int, c_int, c_int, c_double, P, P, P, P, c_int, c_float, c_float, c_float,
                                     P, P, P]),
    "ddsp_hip_sine_source_drawn": (c_int, [P, c_int, c_int, c_int, c_double, P, ctypes.c_ulonglong, ctypes.c_ulonglong, P, P, c_int,
                                           c_float, c_float, c_float, P, P, P]),
    "ddsp_hip_normal_noise": (c_int, [ctypes.c_ulonglong, ctypes.c_ulonglong, c_int, c_long, c_int, P, P]),
    "ddsp_hip_spectral_loss_scratch_bytes": (c_size_t, [c_int, c_long]),
    "ddsp_hip_spectral_loss": (c_int, [P, P, c_int, c_long, c_float, c_float, c_float, P, c_size_t, P, P, P]),
    "ddsp_hip_spectral_loss_backward": (c_int, [P, P, c_int, c_long, P, c_float, c_float, c_float, P, c_int, P, P]),
    "ddsp_hip_stft_loss_table_bytes": (c_size_t, [c_int]),
    "ddsp_hip_stft_loss_tables": (c_int, [c_int, P, P]),
    "ddsp_hip_stft_loss_frames": (c_int, [c_int, c_int, c_int]),
    "ddsp_hip_stft_loss_scratch_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ddsp_hip_stft_loss": (c_int, [P, P, c_int, c_int, c_long, c_int, c_int, P, c_float, c_float, c_float, P, c_size_t,
                                   P, P, P, P, P]),
    "ddsp_hip_stft_loss_backward_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "ddsp_hip_stft_loss_backward": (c_int, [P, P, c_int, c_int, c_int, c_int, P, P, c_float, c_float, c_float, P, c_int, P,
                                            c_long, c_int, P, c_size_t, P]),
    "ddsp_hip_mel_frames": (c_int, [c_int, c_int, c_int]),
    "ddsp_hip_mel_spectrogram": (c_int, [P, c_int, c_int, P, c_int, c_int, P, P, P, c_int, c_int, c_float, P,
                                         c_long, c_long, c_long, P]),
    "ddsp_hip_mel_shifted_table_bytes": (c_size_t, [c_int, c_int]),
    "ddsp_hip_mel_shifted_tables": (c_int, [c_int, c_int, c_int, P, P]),
    "ddsp_hip_mel_shifted_frames": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "ddsp_hip_mel_shifted_spectrogram": (c_int, [P, c_int, c_int, P, c_int, c_int, c_int, c_int, c_int, c_float, P, P, c_int,
                                                 c_float, P, c_long, c_long, c_long, P]),
}

MODE_ROLL, MODE_HANN, MODE_DYNAMIC = 0, 1, 2
ACT_NONE, ACT_EXP = 0, 1
FIR_AUTO, FIR_SIMPLE, FIR_MFMA, FIR_MFMA8, FIR_FFT, FIR_BLK = 0, 1, 2, 3, 4, 5

_LIB = None
_LOCK = threading.Lock()


def bind(cdll):
    """Attach restype/argtypes for every exported entry point; raises if a symbol is missing."""
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)            # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    return cdll


def lib():
    """The loaded libddsp_hip.so.  Built in-tree with hipcc on first use if absent."""
    global _LIB
    if _LIB is None:
        with _LOCK:
            if _LIB is None:
                path = os.environ.get("DDSP_HIP_LIB")          # A/B measurements against another build of the library
                if not path:
                    path = _build.LIB
                    if not os.path.exists(path):
                        path = _build.build()
                _LIB = bind(ctypes.CDLL(path))
    return _LIB


def check_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ddsp_svc_amd: tensors must live on the MI355X (got a %s tensor); "
                               "this package has no CPU path" % t.device.type)


def stream_of(t):
    return torch.cuda.current_stream(t.device).cuda_stream if t.is_cuda else 0


import collections

_AUX = collections.OrderedDict()                              # (device, raw handle of the caller's stream) -> second stream, LRU
_AUX_MAX = 16                                                  # callers that create short-lived streams do not grow it without bound


def _env_flag(name):
    """integer environment switch: unset, empty or 0 = off"""
    v = os.environ.get(name, "").strip()
    try:
        return int(v) != 0 if v else False
    except ValueError:
        return True


_ONE_STREAM = _env_flag("DDSP_HIP_ONE_STREAM")                 # read ONCE, at import: never hand the tails a second stream
try:                                                            # priority of the second stream (A/B switch; see aux_torch_stream)
    _AUX_PRIORITY = int(os.environ.get("DDSP_HIP_AUX_PRIORITY", "0"))
except ValueError:
    _AUX_PRIORITY = 0


def set_tuning(name, value):
    """Measurement / test hook: ``ddsp_hip_set_tuning`` (include/ddsp_hip.h); ``value = 0`` restores the default."""
    check(lib().ddsp_hip_set_tuning(name.encode(), int(value)))


def aux_torch_stream(t, rows):
    """The cached second ``torch.cuda.Stream`` that belongs to the caller's current stream on ``t``'s device (one per
    (device, main stream): callers on different streams -- other host threads -- do not share a branch stream), or None
    (small launches, CPU tensors under the emulator, ``DDSP_HIP_ONE_STREAM=1``, or a device other than the current one)."""
    if not t.is_cuda or rows < 4096 or _ONE_STREAM:
        return None
    if torch.cuda.current_device() != t.device.index:      # the library's fork / join events belong to the current device
        return None
    key = (t.device.index, torch.cuda.current_stream(t.device).cuda_stream)
    with _LOCK:
        s = _AUX.get(key)
        if s is None:
            s = _AUX[key] = torch.cuda.Stream(device=t.device, priority=_AUX_PRIORITY)
            while len(_AUX) > _AUX_MAX:                    # least recently used first; a dropped stream is destroyed by torch
                _AUX.popitem(last=False)                   # once the work already enqueued on it has finished
        else:
            _AUX.move_to_end(key)
    return s


def aux_stream_of(t, rows):
    """Second stream for the noise branch of the synthesiser tails (include/ddsp_hip.h, ``aux_stream``): one per
    device, used once the launch is large enough for the overlap to matter.  ``DDSP_HIP_ONE_STREAM=1`` disables it."""
    s = aux_torch_stream(t, rows)
    return None if s is None else s.cuda_stream


def on_aux_stream(fn, ref, rows, defer_join=False):
    """Run ``fn()`` (a branch that is independent of what the caller's stream does next) on the second stream, forked
    behind the caller's stream and joined back into it; autograd runs the branch's backward on the same stream.
    Returns ``fn``'s tensor -- with ``defer_join`` the pair ``(tensor, join)``: the caller's stream is NOT made to wait
    yet, ``join()`` does that, and the caller invokes it right before the first consumer on its own stream (so the launches
    in between really run beside the branch)."""
    aux = aux_torch_stream(ref, rows)
    if aux is None:
        out = fn()
        return (out, lambda: None) if defer_join else out
    main = torch.cuda.current_stream(ref.device)
    aux.wait_stream(main)
    with torch.cuda.stream(aux):
        out = fn()
    out.record_stream(main)                                # allocated under the second stream, consumed on the first
    if defer_join:
        return out, (lambda: main.wait_stream(aux))
    main.wait_stream(aux)
    return out


def ptr(t):
    return None if t is None else t.data_ptr()


def check(code):
    if code != 0:
        msg = lib().ddsp_hip_error_string(code)
        raise RuntimeError("libddsp_hip: %s (code %d)" % (msg.decode() if msg else "?", code))
