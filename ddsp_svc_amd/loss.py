"""Drop-in for the training loss of the reference, ``ddsp/loss.py``: ``SSSLoss`` (:9-32) and ``RSSLoss`` (:34-54) --
same constructors, same ``forward`` argument order (``SSSLoss(x_true, x_pred)``, ``RSSLoss(x_pred, x_true)``), same
random draw of the transform sizes (``torch.randint``, :47).

``RSSLoss`` draws arbitrary integer transform sizes, most of them with large prime factors.  The whole loss runs in
csrc/loss_czt.hip straight from the two waveforms: one chirp-z transform per frame carries both signals, the magnitudes
and reductions follow in registers, and the backward pass is one more transform per two frames (``_WaveLossFunction``;
all scales of ``RSSLoss`` as one autograd node, ``_RandomScaleWaveLossFunction``).  With ``overlap = 0`` (the default,
and what ``train.py`` uses) the frames' gradients are the signal's; with overlapping frames a second kernel gathers
them per sample.  ``torch.stft`` + the fused reductions of csrc/loss.hip (``_SpectralLossFunction``) remain for what the
plans do not cover -- transform sizes above 2048 -- and behind ``DDSP_HIP_LOSS_TORCH_STFT=1`` for A/B measurements."""
import torch

from . import _ffi
from ._ffi import ptr


def _dense_batch_major(z):
    """True when ``z`` ([B, bins, frames] complex) occupies one dense block per utterance, utterances in order."""
    if z.dim() != 3:
        return False
    per = z.shape[1] * z.shape[2]
    inner = sorted(zip(z.stride()[1:], z.shape[1:]))
    dense = inner[0][0] == 1 and inner[1][0] == inner[0][1]
    return dense and (z.shape[0] == 1 or z.stride(0) == per)


class _SpectralLossFunction(torch.autograd.Function):
    """loss.py:22-31 behind the two STFTs."""

    @staticmethod
    def forward(ctx, spec_true, spec_pred, inv_window_norm, eps, alpha):
        _ffi.check_device(spec_true, spec_pred)
        if spec_true.shape != spec_pred.shape or spec_true.dim() != 3:
            raise ValueError("the two spectrograms must have the same [B, bins, frames] shape")
        st, sp = spec_true.detach(), spec_pred.detach()
        if st.dtype != torch.complex64 or sp.dtype != torch.complex64:
            st, sp = st.to(torch.complex64), sp.to(torch.complex64)
        if not (_dense_batch_major(st) and st.stride() == sp.stride()):
            st, sp = st.contiguous(), sp.contiguous()
        B = st.shape[0]
        per = st.shape[1] * st.shape[2]
        lib = _ffi.lib()
        dev = st.device
        nbytes = lib.ddsp_hip_spectral_loss_scratch_bytes(B, per)
        scratch = torch.empty(max(nbytes, 8) // 8, dtype=torch.float64, device=dev)
        norms = torch.empty(B, 2, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _ffi.check(lib.ddsp_hip_spectral_loss(ptr(st), ptr(sp), B, per, float(inv_window_norm), float(eps), float(alpha),
                                              ptr(scratch), nbytes, ptr(norms), ptr(loss), _ffi.stream_of(st)))
        ctx.save_for_backward(st, sp, norms)
        ctx.cfg = (float(inv_window_norm), float(eps), float(alpha))
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        st, sp, norms = ctx.saved_tensors
        inv_wn, eps, alpha = ctx.cfg
        B = st.shape[0]
        per = st.shape[1] * st.shape[2]
        go = grad_out.detach().to(torch.float32).contiguous()
        lib = _ffi.lib()
        grads = [None, None]
        for which in (0, 1):                                         # 0: true, 1: pred
            if not ctx.needs_input_grad[which]:
                continue
            d = torch.empty_like(sp)                                 # same dense layout as the saved spectra
            _ffi.check(lib.ddsp_hip_spectral_loss_backward(ptr(st), ptr(sp), B, per, ptr(norms), inv_wn, eps, alpha,
                                                           ptr(go), 1 if which == 0 else 0, ptr(d),
                                                           _ffi.stream_of(sp)))
            grads[which] = d
        return grads[0], grads[1], None, None, None


import collections
import threading

_CZT_TABLES = collections.OrderedDict()    # (device, stream, n_fft) -> chirps and filter spectrum of csrc/loss_czt.hip, LRU
_CZT_TABLES_LOCK = threading.Lock()        # host threads of one process share the cache (get / insert / evict are not atomic)
_CZT_TABLES_BYTES = 64 << 20               # byte budget of the cache: every size RSSLoss draws from [256, 2048) is ~47 MB per stream
_czt_tables_held = 0


def _czt_tables(n_fft, like):
    """The tables of one transform size on ``like``'s device, filled on first use (on the current stream, which is part
    of the key: a second stream gets its own copy rather than a race with the fill).  None: size not supported.
    A table first needed while a HIP graph is being captured is not cached: its contents exist only after a replay."""
    global _czt_tables_held
    key = (str(like.device), _ffi.stream_of(like), int(n_fft))
    with _CZT_TABLES_LOCK:
        t = _CZT_TABLES.get(key)
        if t is not None:
            _CZT_TABLES.move_to_end(key)
            return t
    lib = _ffi.lib()
    nbytes = lib.ddsp_hip_stft_loss_table_bytes(int(n_fft))
    if nbytes == 0:
        return None
    t = torch.empty(nbytes // 4, dtype=torch.float32, device=like.device)
    _ffi.check(lib.ddsp_hip_stft_loss_tables(int(n_fft), ptr(t), _ffi.stream_of(like)))
    if like.is_cuda and torch.cuda.is_current_stream_capturing():
        return t
    with _CZT_TABLES_LOCK:
        other = _CZT_TABLES.get(key)
        if other is not None:                                  # another thread filled the same key meanwhile: one copy is kept
            return other
        _CZT_TABLES[key] = t
        _czt_tables_held += nbytes
        while _czt_tables_held > _CZT_TABLES_BYTES and len(_CZT_TABLES) > 1:
            _, dropped = _CZT_TABLES.popitem(last=False)       # (freed once the work already enqueued on it is done)
            _czt_tables_held -= dropped.numel() * 4
    return t


class _WaveLossFunction(torch.autograd.Function):
    """loss.py:22-31 INCLUDING the two spectrograms (csrc/loss_czt.hip), any 1 <= hop <= n_fft."""

    @staticmethod
    def forward(ctx, x_true, x_pred, n_fft, hop, inv_window_norm, eps, alpha, tables):
        _ffi.check_device(x_true, x_pred)
        xt, xp = x_true.detach(), x_pred.detach()
        if xt.stride(1) != 1 or xp.stride(1) != 1 or xt.stride(0) != xp.stride(0):
            xt, xp = xt.contiguous(), xp.contiguous()
        B, T = xt.shape
        lib = _ffi.lib()
        dev = xt.device
        frames = lib.ddsp_hip_stft_loss_frames(T, n_fft, hop)
        spec = torch.empty(2, B, frames, n_fft // 2 + 1, dtype=torch.complex64, device=dev)
        nbytes = lib.ddsp_hip_stft_loss_scratch_bytes(B, T, n_fft, hop)
        scratch = torch.empty(max(nbytes, 8) // 8, dtype=torch.float64, device=dev)
        norms = torch.empty(B, 2, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _ffi.check(lib.ddsp_hip_stft_loss(ptr(xt), ptr(xp), B, T, xt.stride(0), n_fft, hop, ptr(tables),
                                          float(inv_window_norm), float(eps), float(alpha), ptr(scratch), nbytes,
                                          ptr(spec[0]), ptr(spec[1]), ptr(norms), ptr(loss), _ffi.stream_of(xt)))
        ctx.save_for_backward(spec, norms, tables)
        ctx.cfg = (B, T, int(n_fft), int(hop), float(inv_window_norm), float(eps), float(alpha))
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        spec, norms, tables = ctx.saved_tensors
        B, T, n_fft, hop, inv_wn, eps, alpha = ctx.cfg
        go = grad_out.detach().to(torch.float32).contiguous()
        lib = _ffi.lib()
        ws_bytes = lib.ddsp_hip_stft_loss_backward_ws_bytes(B, T, n_fft, hop)      # overlapping frames: their gradients, then a gather
        ws = torch.empty(max(ws_bytes, 16) // 4, dtype=torch.float32, device=spec.device)
        grads = [None, None]
        for which in (0, 1):                                         # 0: true, 1: pred
            if not ctx.needs_input_grad[which]:
                continue
            d = torch.empty(B, T, dtype=torch.float32, device=spec.device)
            _ffi.check(lib.ddsp_hip_stft_loss_backward(ptr(spec[0]), ptr(spec[1]), B, T, n_fft, hop, ptr(tables), ptr(norms),
                                                       inv_wn, eps, alpha, ptr(go), 1 if which == 0 else 0, ptr(d), T, 0,
                                                       ptr(ws), ws_bytes, _ffi.stream_of(spec)))
            grads[which] = d
        return grads[0], grads[1], None, None, None, None, None, None


class _RandomScaleWaveLossFunction(torch.autograd.Function):
    """RSSLoss.forward (loss.py:46-54) as ONE autograd node: the scales' kernels back to back,
    their losses averaged on the device, and in the backward pass each scale's kernel adding into the one gradient
    buffer -- instead of n_scale nodes, n_scale [B, T] temporaries and the adds between them."""

    @staticmethod
    def forward(ctx, x_true, x_pred, scales, eps, alpha, *tables):
        _ffi.check_device(x_true, x_pred)
        xt, xp = x_true.detach(), x_pred.detach()
        if xt.stride(1) != 1 or xp.stride(1) != 1 or xt.stride(0) != xp.stride(0):
            xt, xp = xt.contiguous(), xp.contiguous()
        B, T = xt.shape
        lib = _ffi.lib()
        dev = xt.device
        losses = torch.empty(len(scales), dtype=torch.float32, device=dev)
        norms = torch.empty(len(scales), B, 2, dtype=torch.float32, device=dev)
        nbytes = max(lib.ddsp_hip_stft_loss_scratch_bytes(B, T, n, hop) for n, hop, _ in scales)
        scratch = torch.empty(max(nbytes, 8) // 8, dtype=torch.float64, device=dev)   # the scales run in stream order
        specs = []
        for i, ((n, hop, inv_wn), tab) in enumerate(zip(scales, tables)):
            frames = lib.ddsp_hip_stft_loss_frames(T, n, hop)
            spec = torch.empty(2, B, frames, n // 2 + 1, dtype=torch.complex64, device=dev)
            _ffi.check(lib.ddsp_hip_stft_loss(ptr(xt), ptr(xp), B, T, xt.stride(0), n, hop, ptr(tab), inv_wn, float(eps),
                                              float(alpha), ptr(scratch), nbytes, ptr(spec[0]), ptr(spec[1]),
                                              ptr(norms[i]), ptr(losses[i:]), _ffi.stream_of(xt)))
            specs.append(spec)
        ctx.save_for_backward(norms, *specs, *tables)
        ctx.cfg = (B, T, tuple(scales), float(eps), float(alpha))
        return losses.sum() / len(scales)                             # loss.py:48-54: the sum over the scales / n_scale

    @staticmethod
    def backward(ctx, grad_out):
        B, T, scales, eps, alpha = ctx.cfg
        k = len(scales)
        norms, specs, tables = ctx.saved_tensors[0], ctx.saved_tensors[1:1 + k], ctx.saved_tensors[1 + k:]
        go = (grad_out.detach().to(torch.float32) / k).contiguous()
        lib = _ffi.lib()
        ws_bytes = max(lib.ddsp_hip_stft_loss_backward_ws_bytes(B, T, n, hop) for n, hop, _ in scales)
        ws = torch.empty(max(ws_bytes, 16) // 4, dtype=torch.float32, device=norms.device)
        grads = [None, None]
        for which in (0, 1):                                         # 0: true, 1: pred
            if not ctx.needs_input_grad[which]:
                continue
            d = torch.empty(B, T, dtype=torch.float32, device=norms.device)
            for i, ((n, hop, inv_wn), spec, tab) in enumerate(zip(scales, specs, tables)):
                _ffi.check(lib.ddsp_hip_stft_loss_backward(ptr(spec[0]), ptr(spec[1]), B, T, n, hop, ptr(tab), ptr(norms[i]),
                                                           inv_wn, eps, alpha, ptr(go), 1 if which == 0 else 0, ptr(d), T,
                                                           1 if i else 0, ptr(ws), ws_bytes, _ffi.stream_of(norms)))
            grads[which] = d
        return (grads[0], grads[1], None, None, None) + (None,) * k


class Spectrogram(torch.nn.Module):
    """``torchaudio.transforms.Spectrogram(n_fft, hop_length=, power=None, center=False)`` as SSSLoss configures it
    (loss.py:20), returning the COMPLEX spectrum: the ``power=1`` magnitude and the ``normalized=True`` division by
    ``||window||_2`` are applied inside the fused loss kernel."""

    def __init__(self, n_fft, hop_length):
        super().__init__()
        self.n_fft = int(n_fft)
        self.hop_length = int(hop_length)
        window = torch.hann_window(self.n_fft)                              # periodic Hann, torchaudio's default
        self.register_buffer("window", window)
        self.inv_window_norm = 1.0 / float(window.double().pow(2).sum().sqrt())     # a host float: no device sync per call

    def forward(self, x):
        x = x.reshape(-1, x.shape[-1])
        n, hop = self.n_fft, self.hop_length
        if hop == n and x.shape[-1] >= n:
            # overlap = 0 (loss.py:14, the configuration train.py uses): the frames do not overlap, so the STFT is a
            # view, one window multiply and one batched rfft -- [B, frames, bins], which the loss kernels take as is
            # (they only need one dense block per utterance) -- and its backward the matching c2r transform instead
            # of torch.stft's generic unfold / fold pair
            frames = x.shape[-1] // n
            return torch.fft.rfft(x[:, :frames * n].reshape(x.shape[0], frames, n) * self.window, dim=-1)
        return torch.stft(x, n, hop_length=hop, win_length=n, window=self.window, center=False, normalized=False,
                          onesided=True, return_complex=True)


class SSSLoss(torch.nn.Module):
    """Single-scale spectral loss, loss.py:9-32."""

    def __init__(self, n_fft=111, alpha=1.0, overlap=0, eps=1e-7):
        super().__init__()
        self.n_fft = n_fft
        self.alpha = alpha
        self.eps = eps
        self.hop_length = int(n_fft * (1 - overlap))                 # loss.py:19
        self.spec = Spectrogram(self.n_fft, self.hop_length)

    def forward(self, x_true, x_pred):
        _ffi.check_device(x_true, x_pred)
        if x_true.shape != x_pred.shape:
            raise ValueError("x_true and x_pred must have the same shape")
        x_true = x_true.reshape(-1, x_true.shape[-1]) if x_true.dim() != 2 else x_true
        x_pred = x_pred.reshape(-1, x_pred.shape[-1]) if x_pred.dim() != 2 else x_pred
        if 1 <= self.hop_length <= self.n_fft and x_true.shape[-1] >= self.n_fft and x_true.shape[-1] < 2 ** 31 \
                and x_true.shape[0] <= 65535 and not _ffi._env_flag("DDSP_HIP_LOSS_TORCH_STFT"):
            tables = _czt_tables(self.n_fft, x_pred)
            if tables is not None:
                return _WaveLossFunction.apply(x_true.to(torch.float32), x_pred.to(torch.float32), self.n_fft,
                                               self.hop_length, self.spec.inv_window_norm, self.eps, self.alpha, tables)
        return _SpectralLossFunction.apply(self.spec(x_true.to(torch.float32)), self.spec(x_pred.to(torch.float32)),
                                           self.spec.inv_window_norm, self.eps, self.alpha)


class RSSLoss(torch.nn.Module):
    """Random-scale spectral loss, loss.py:34-54.  The per-size modules are built on first use rather than all
    ``fft_max - fft_min`` of them up front (loss.py:44-45 holds ~1800 window buffers); ``lossdict`` fills as sizes are
    drawn."""

    def __init__(self, fft_min, fft_max, n_scale, alpha=1.0, overlap=0, eps=1e-7, device="cuda"):
        super().__init__()
        self.fft_min = fft_min
        self.fft_max = fft_max
        self.n_scale = n_scale
        self.alpha, self.overlap, self.eps, self.device = alpha, overlap, eps, device
        self.lossdict = {}

    def _scale(self, n_fft, device):
        f = self.lossdict.get(n_fft)
        if f is None:
            f = self.lossdict[n_fft] = SSSLoss(n_fft, self.alpha, self.overlap, self.eps).to(device)
        return f

    def _fused(self, x_pred, x_true, sizes):
        """All scales in one autograd node when every one of them takes the in-kernel transform (sizes the chirp-z plans
        cover, hops of at least one sample, signals of at least one frame); None otherwise."""
        # ascending sizes: scales that share a transform plan (1024 / 2048 / 4096 points) then launch back to back, and the
        # second launch finds the kernel's code in the instruction cache (the sum over the scales does not care about order)
        sizes = sorted(sizes)
        hops = [int(n * (1 - self.overlap)) for n in sizes]                      # loss.py:19
        if any(h < 1 or h > n for h, n in zip(hops, sizes)):
            return None
        if _ffi._env_flag("DDSP_HIP_LOSS_TORCH_STFT") or x_true.shape != x_pred.shape:
            return None
        T = x_true.shape[-1]
        if T < max(sizes) or T >= 2 ** 31 or len(sizes) == 0 or x_true.numel() // T > 65535:
            return None
        _ffi.check_device(x_true, x_pred)
        xt = x_true.reshape(-1, T).to(torch.float32)
        xp = x_pred.reshape(-1, T).to(torch.float32)
        tables = [_czt_tables(n, xp) for n in sizes]
        if any(t is None for t in tables):
            return None
        scales = tuple((n, h, self._scale(n, xp.device).spec.inv_window_norm) for n, h in zip(sizes, hops))
        return _RandomScaleWaveLossFunction.apply(xt, xp, scales, self.eps, self.alpha, *tables)

    def forward(self, x_pred, x_true):
        value = 0.
        n_ffts = torch.randint(self.fft_min, self.fft_max, (self.n_scale,))      # loss.py:47, CPU generator
        fused = self._fused(x_pred, x_true, [int(n) for n in n_ffts])
        if fused is not None:
            return fused
        for n_fft in n_ffts:
            value = value + self._scale(int(n_fft), x_pred.device)(x_true, x_pred)
        return value / self.n_scale


def patch_reference_loss():
    """Rebind ``SSSLoss`` / ``RSSLoss`` of an importable reference checkout (``ddsp.loss`` and the name ``train.py``
    imported at :7) to the classes above."""
    import sys
    import ddsp.loss as dl
    if not hasattr(dl, "_reference_RSSLoss"):
        dl._reference_SSSLoss, dl._reference_RSSLoss = dl.SSSLoss, dl.RSSLoss
    dl.SSSLoss, dl.RSSLoss = SSSLoss, RSSLoss
    tr = sys.modules.get("train")
    if tr is not None and hasattr(tr, "RSSLoss"):
        tr.RSSLoss = RSSLoss
    return dl
