"""Drop-in ``Sins`` / ``CombSub`` / ``CombSubFast`` / ``CombSubSuperFast`` modules (reference:
ddsp/vocoder.py:532-611, :788-862, :712-786, :613-710).

Same constructor arguments, buffers (``sampling_rate``, ``block_size``), child module name
(``unit2ctrl``) and ``forward`` signature/return value as the reference, so checkpoints load
with ``strict=True`` and ``load_model`` / the diffusion cascade / the enhancer keep working.
``Unit2Control`` is NOT re-implemented: it is the reference's own PyTorch module (north star:
"keeps working unmodified"), imported from the user's DDSP-SVC checkout, or injected through
``unit2ctrl_factory``.  Only the DSP around it runs on the HIP kernels.
"""
import torch

from . import synth


def _reference_unit2control():
    try:
        from ddsp.unit2control import Unit2Control       # the DDSP-SVC checkout must be on sys.path
    except Exception as e:                                # pragma: no cover - message path
        raise ImportError("ddsp_svc_amd.vocoder needs the reference's ddsp.unit2control.Unit2Control "
                          "(put the DDSP-SVC checkout on sys.path) or an explicit unit2ctrl_factory") from e
    return Unit2Control


class _SynthBase(torch.nn.Module):
    # patch_reference() stores the reference's class of the same name here: tensors that are not on the GPU are handed to
    # ITS forward (the reference's own CPU code, e.g. load_model's default device='cpu', vocoder.py:506) -- this package
    # has no CPU arithmetic of its own.  Without a patched-in reference a host tensor is an error (check_device).
    _reference_cls = None

    def _to_reference(self, f0_frames):
        return self._reference_cls is not None and not f0_frames.is_cuda

    def __init__(self, sampling_rate, block_size, split_map, n_unit, n_spk, unit2ctrl_factory=None,
                 **unit2ctrl_kwargs):
        super().__init__()
        # 0-dim buffers exactly as the reference registers them (state_dict compatibility); cached
        # Python numbers avoid the .item() device syncs the reference pays on every forward
        self.register_buffer("sampling_rate", torch.tensor(sampling_rate))
        self.register_buffer("block_size", torch.tensor(block_size))
        self._sr = float(sampling_rate)
        self._hop = int(block_size)
        factory = unit2ctrl_factory or _reference_unit2control()
        self.unit2ctrl = factory(n_unit, n_spk, split_map, **unit2ctrl_kwargs)
        self.return_components = True       # the (harmonic, noise) tuple is API; set False to skip materialising it
        self.fir_impl = 0
        # Opt-in (Sins / CombSub): an integer seed makes the noise draw happen INSIDE the noise filter (a Philox stream
        # of its own, synth.uniform_noise -- not the numbers torch.rand_like would give) instead of a torch.rand tensor;
        # every forward call advances the stream (offset = number of calls so far).  None: torch.rand as the reference.
        self.in_kernel_noise_seed = None
        self._noise_calls = 0

    def _noise(self, B, T, device):
        """(noise tensor | None, kwargs) for the synth call"""
        if self.in_kernel_noise_seed is None:
            return torch.rand(B, T, dtype=torch.float32, device=device), {}
        self._noise_calls += 1
        return None, {"noise_seed": int(self.in_kernel_noise_seed), "noise_offset": self._noise_calls - 1}

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._sr = float(self.sampling_rate)
        self._hop = int(self.block_size)


class Sins(_SynthBase):
    """Sinusoids additive synthesiser, ddsp/vocoder.py:532-611."""

    def __init__(self, sampling_rate, block_size, n_harmonics, n_mag_allpass, n_mag_noise, n_unit=256, n_spk=1,
                 unit2ctrl_factory=None):
        split_map = {"amplitudes": n_harmonics, "group_delay": n_mag_allpass, "noise_magnitude": n_mag_noise}
        super().__init__(sampling_rate, block_size, split_map, n_unit, n_spk, unit2ctrl_factory)

    def forward(self, units_frames, f0_frames, volume_frames, spk_id=None, spk_mix_dict=None, initial_phase=None,
                infer=True, max_upsample_dim=32):
        if self._to_reference(f0_frames):
            return self._reference_cls.forward(self, units_frames, f0_frames, volume_frames, spk_id=spk_id,
                                               spk_mix_dict=spk_mix_dict, initial_phase=initial_phase, infer=infer,
                                               max_upsample_dim=max_upsample_dim)
        st = synth.phase(f0_frames, self._sr, self._hop, initial_phase, infer)                 # :564-575
        ctrls, hidden = self.unit2ctrl(units_frames, f0_frames, st.phase_frames, volume_frames,
                                       spk_id=spk_id, spk_mix_dict=spk_mix_dict)                # :578
        B, F = f0_frames.shape[0], f0_frames.shape[1]
        u01, rng = self._noise(B, F * self._hop, f0_frames.device)                               # rand_like, :603
        signal, harmonic, noise = synth.sins_synth(
            f0_frames, st, ctrls["amplitudes"], ctrls["group_delay"], ctrls["noise_magnitude"], u01,
            self._sr, self._hop, noise_is_u01=True, want_components=self.return_components,
            fir_impl=self.fir_impl, **rng)
        return signal, hidden, (harmonic, noise)


class CombSub(_SynthBase):
    """Combtooth subtractive synthesiser (old version), ddsp/vocoder.py:788-862."""

    def __init__(self, sampling_rate, block_size, n_mag_allpass, n_mag_harmonic, n_mag_noise, n_unit=256, n_spk=1,
                 unit2ctrl_factory=None):
        split_map = {"group_delay": n_mag_allpass, "harmonic_magnitude": n_mag_harmonic,
                     "noise_magnitude": n_mag_noise}
        super().__init__(sampling_rate, block_size, split_map, n_unit, n_spk, unit2ctrl_factory)

    def forward(self, units_frames, f0_frames, volume_frames, spk_id=None, spk_mix_dict=None, initial_phase=None,
                infer=True, **kwargs):
        if self._to_reference(f0_frames):
            return self._reference_cls.forward(self, units_frames, f0_frames, volume_frames, spk_id=spk_id,
                                               spk_mix_dict=spk_mix_dict, initial_phase=initial_phase, infer=infer, **kwargs)
        st = synth.phase(f0_frames, self._sr, self._hop, initial_phase, infer)                 # :819-829
        ctrls, hidden = self.unit2ctrl(units_frames, f0_frames, st.phase_frames, volume_frames,
                                       spk_id=spk_id, spk_mix_dict=spk_mix_dict)                # :832
        B, F = f0_frames.shape[0], f0_frames.shape[1]
        u01, rng = self._noise(B, F * self._hop, f0_frames.device)                               # rand_like, :854
        signal, harmonic, noise = synth.combsub_synth(
            f0_frames, st, ctrls["group_delay"], ctrls["harmonic_magnitude"], ctrls["noise_magnitude"], u01,
            self._sr, self._hop, noise_is_u01=True, want_components=self.return_components,
            fir_impl=self.fir_impl, **rng)
        return signal, hidden, (harmonic, noise)


class CombSubFast(_SynthBase):
    """Combtooth subtractive synthesiser, ddsp/vocoder.py:712-786: sqrt-Hann frames of ``2*block_size``,
    per-frame complex source filter and zero-phase noise filter in the rfft domain, overlap-add."""

    def __init__(self, sampling_rate, block_size, n_unit=256, n_spk=1, use_pitch_aug=False, pcmer_norm=False,
                 unit2ctrl_factory=None):
        split_map = {"harmonic_magnitude": block_size + 1, "harmonic_phase": block_size + 1,
                     "noise_magnitude": block_size + 1}                                          # :728-732
        super().__init__(sampling_rate, block_size, split_map, n_unit, n_spk, unit2ctrl_factory,
                         use_pitch_aug=use_pitch_aug, pcmer_norm=pcmer_norm)                     # :733
        self.register_buffer("window", torch.sqrt(torch.hann_window(2 * block_size)))           # :726

    def forward(self, units_frames, f0_frames, volume_frames, spk_id=None, spk_mix_dict=None, aug_shift=None,
                initial_phase=None, infer=True, **kwargs):
        if self._to_reference(f0_frames):
            return self._reference_cls.forward(self, units_frames, f0_frames, volume_frames, spk_id=spk_id,
                                               spk_mix_dict=spk_mix_dict, aug_shift=aug_shift,
                                               initial_phase=initial_phase, infer=infer, **kwargs)
        st = synth.phase(f0_frames, self._sr, self._hop, initial_phase, infer)                 # :743-753
        ctrls, hidden = self.unit2ctrl(units_frames, f0_frames, st.phase_frames, volume_frames,
                                       spk_id=spk_id, spk_mix_dict=spk_mix_dict, aug_shift=aug_shift)   # :756
        B, F = f0_frames.shape[0], f0_frames.shape[1]
        u01 = torch.rand(B, F * self._hop, dtype=torch.float32, device=f0_frames.device)        # rand_like, :771
        signal = synth.combsubfast_synth(f0_frames, st, ctrls["harmonic_magnitude"], ctrls["harmonic_phase"],
                                         ctrls["noise_magnitude"], u01, self.window, self._sr, self._hop,
                                         noise_is_u01=True)
        return signal, hidden, (signal, signal)                                                 # :786


class CombSubSuperFast(_SynthBase):
    """Combtooth subtractive synthesiser, ddsp/vocoder.py:613-710 (the model ``configs/combsub.yaml`` ships and
    the DDSP stage of the diffusion / reflow cascades): closed-form exciter phase, ``torch.stft`` /
    ``torch.istft`` framing with a Hann window of ``win_length``, complex source and noise filters."""

    def __init__(self, sampling_rate, block_size, win_length, n_unit=256, n_spk=1, use_pitch_aug=False,
                 pcmer_norm=False, unit2ctrl_factory=None):
        n = win_length // 2 + 1
        split_map = {"harmonic_magnitude": n, "harmonic_phase": n, "noise_magnitude": n, "noise_phase": n}   # :631-636
        super().__init__(sampling_rate, block_size, split_map, n_unit, n_spk, unit2ctrl_factory,
                         use_pitch_aug=use_pitch_aug, use_naive_v2=True, use_conv_stack=True)    # :637
        self.register_buffer("win_length", torch.tensor(win_length))                            # :628
        self.register_buffer("window", torch.hann_window(win_length))                           # :629

    def fast_source_gen(self, f0_frames):
        """vocoder.py:639-651 -> ``(combtooth [B,T], phase_frames [B,F,1])``."""
        if self._to_reference(f0_frames):
            return self._reference_cls.fast_source_gen(self, f0_frames)
        st = synth.fast_source(f0_frames, self._sr, self._hop, want_combtooth=True)
        return st.combtooth, st.phase_frames

    def forward(self, units_frames, f0_frames, volume_frames, spk_id=None, spk_mix_dict=None, aug_shift=None,
                initial_phase=None, infer=True, **kwargs):
        if self._to_reference(f0_frames):
            return self._reference_cls.forward(self, units_frames, f0_frames, volume_frames, spk_id=spk_id,
                                               spk_mix_dict=spk_mix_dict, aug_shift=aug_shift,
                                               initial_phase=initial_phase, infer=infer, **kwargs)
        st = synth.fast_source(f0_frames, self._sr, self._hop)                                  # :653
        ctrls, hidden = self.unit2ctrl(units_frames, f0_frames, st.phase_frames, volume_frames,
                                       spk_id=spk_id, spk_mix_dict=spk_mix_dict, aug_shift=aug_shift)   # :656
        B, F = f0_frames.shape[0], f0_frames.shape[1]
        gauss = torch.randn(B, F * self._hop, dtype=torch.float32, device=f0_frames.device)     # randn_like, :687
        signal = synth.combsubsuperfast_synth(f0_frames, st, ctrls["harmonic_magnitude"], ctrls["harmonic_phase"],
                                              ctrls["noise_magnitude"], ctrls["noise_phase"], gauss, self.window,
                                              self._sr, self._hop)
        return signal, hidden, (signal, signal)                                                 # :710


PATCHED_CORE_FUNCTIONS = ("upsample", "remove_above_fmax", "frequency_filter", "fft_convolve", "frequency_impulse_response",
                          "apply_window_to_impulse_response", "apply_dynamic_window_to_impulse_response")


def _tensors(args, kwargs):
    for v in list(args) + list(kwargs.values()):
        if torch.is_tensor(v):
            yield v


def _rebind_everywhere(mapping, skip=()):
    """Every module that bound one of ``mapping``'s keys by name (``from ddsp.core import upsample`` in flask_api.py:12,
    gui_diff.py:10, main_reflow.py:13, ...; ``from ddsp.vocoder import CombSubFast`` in diffusion/vocoder.py:13) gets the value
    instead -- found by IDENTITY over ``sys.modules``, so no list of importers has to be kept up to date.  Attributes whose name
    starts with ``_reference_`` (where the originals are parked) and the modules in ``skip`` are left alone.  Returns the
    bindings it changed as ``(module, name, old value, new value)`` -- ``unpatch_reference`` undoes exactly those."""
    import sys
    ids = {id(k): v for k, v in mapping.items()}
    changed = []
    for mod in list(sys.modules.values()):
        if mod is None or mod in skip:
            continue
        try:
            items = list(vars(mod).items())
        except TypeError:                                # a module-like object without a __dict__
            continue
        for name, val in items:
            new = ids.get(id(val))
            if new is not None and not name.startswith("_reference_"):
                try:
                    setattr(mod, name, new)
                    changed.append((mod, name, val, new))
                except Exception:                        # noqa: BLE001  (read-only module objects)
                    pass
    return changed


_REBOUND = []                                            # what patch_reference() rebound outside ddsp.core / ddsp.vocoder


def patch_reference():
    """Swap these classes (and the ddsp.core functions on the path) into an already-importable
    reference checkout so ``ddsp.vocoder.load_model``, ``main.py``, ``main_diff.py`` ... pick them up
    without edits.  Modules that bound the names at import time (``from ddsp.core import upsample``: flask_api.py:12,
    flask_api_diff.py:12, gui_diff.py:10, gui_reflow.py:8, main_reflow.py:13, ...; the cascades' ``CombSubFast`` /
    ``CombSubSuperFast``: diffusion/vocoder.py:13, reflow/vocoder.py:12) are rebound wherever they are in ``sys.modules``,
    by identity of the original object, so the call may come before or after those imports.

    Dispatch rule of the patched ``ddsp.core`` functions: GPU tensors go to the HIP kernels (all of them are
    differentiable where the reference's are -- ``upsample`` / ``remove_above_fmax`` through small autograd functions,
    the filters through the adjoint kernels); anything else -- host tensors, dtypes other than float32 / complex64 --
    keeps the reference's own code.  The module classes route host tensors to the reference's ``forward``."""
    import ddsp.core as rcore
    import ddsp.vocoder as rvoc
    from . import core as hcore
    mine = {"Sins": Sins, "CombSub": CombSub, "CombSubFast": CombSubFast, "CombSubSuperFast": CombSubSuperFast}
    swapped = {}
    for name, cls in mine.items():
        if not hasattr(rvoc, "_reference_" + name):
            setattr(rvoc, "_reference_" + name, getattr(rvoc, name))
        cls._reference_cls = getattr(rvoc, "_reference_" + name)
        setattr(rvoc, name, cls)
        swapped[cls._reference_cls] = cls
    for name in PATCHED_CORE_FUNCTIONS:
        if not hasattr(rcore, name):
            continue
        if not hasattr(rcore, "_reference_" + name):
            setattr(rcore, "_reference_" + name, getattr(rcore, name))

            def dispatch(*a, __h=getattr(hcore, name), __r=getattr(rcore, name), **k):
                ts = list(_tensors(a, k))
                on_gpu = bool(ts) and all(t.is_cuda for t in ts)
                plain = all(t.dtype in (torch.float32, torch.complex64) for t in ts)
                return __h(*a, **k) if on_gpu and plain else __r(*a, **k)
            dispatch.__name__ = name
            setattr(rcore, name, dispatch)
        swapped[getattr(rcore, "_reference_" + name)] = getattr(rcore, name)
    _REBOUND.extend(_rebind_everywhere(swapped))
    return rvoc


def unpatch_reference():
    """Undo ``patch_reference()``: the reference's own classes and functions are bound again everywhere."""
    import ddsp.core as rcore
    import ddsp.vocoder as rvoc
    for name, cls in (("Sins", Sins), ("CombSub", CombSub), ("CombSubFast", CombSubFast), ("CombSubSuperFast", CombSubSuperFast)):
        ref = getattr(rvoc, "_reference_" + name, None)
        if ref is None:
            continue
        setattr(rvoc, name, ref)
        delattr(rvoc, "_reference_" + name)
        cls._reference_cls = None
    for name in PATCHED_CORE_FUNCTIONS:
        ref = getattr(rcore, "_reference_" + name, None)
        if ref is None:
            continue
        setattr(rcore, name, ref)
        delattr(rcore, "_reference_" + name)
    while _REBOUND:                                      # exactly the bindings patch_reference() changed, if they are still its
        mod, name, old, new = _REBOUND.pop()
        if getattr(mod, name, None) is new:
            setattr(mod, name, old)
