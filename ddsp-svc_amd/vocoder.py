"""Drop-in ``Sins`` / ``CombSub`` modules (reference: ddsp/vocoder.py:532-611 and :788-862).

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
    def __init__(self, sampling_rate, block_size, split_map, n_unit, n_spk, unit2ctrl_factory=None):
        super().__init__()
        # 0-dim buffers exactly as the reference registers them (state_dict compatibility); cached
        # Python numbers avoid the .item() device syncs the reference pays on every forward
        self.register_buffer("sampling_rate", torch.tensor(sampling_rate))
        self.register_buffer("block_size", torch.tensor(block_size))
        self._sr = float(sampling_rate)
        self._hop = int(block_size)
        factory = unit2ctrl_factory or _reference_unit2control()
        self.unit2ctrl = factory(n_unit, n_spk, split_map)
        self.return_components = True       # the (harmonic, noise) tuple is API; set False to skip materialising it
        self.fir_impl = 0

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._sr = float(self.sampling_rate)
        self._hop = int(self.block_size)

    def _check_inference_only(self, *tensors):
        if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
            raise NotImplementedError("ddsp_svc_amd synthesisers are forward-only (no autograd through the "
                                      "HIP kernels yet); call under torch.no_grad()")


class Sins(_SynthBase):
    """Sinusoids additive synthesiser, ddsp/vocoder.py:532-611."""

    def __init__(self, sampling_rate, block_size, n_harmonics, n_mag_allpass, n_mag_noise, n_unit=256, n_spk=1,
                 unit2ctrl_factory=None):
        split_map = {"amplitudes": n_harmonics, "group_delay": n_mag_allpass, "noise_magnitude": n_mag_noise}
        super().__init__(sampling_rate, block_size, split_map, n_unit, n_spk, unit2ctrl_factory)

    def forward(self, units_frames, f0_frames, volume_frames, spk_id=None, spk_mix_dict=None, initial_phase=None,
                infer=True, max_upsample_dim=32):
        st = synth.phase(f0_frames, self._sr, self._hop, initial_phase, infer)                 # :564-575
        ctrls, hidden = self.unit2ctrl(units_frames, f0_frames, st.phase_frames, volume_frames,
                                       spk_id=spk_id, spk_mix_dict=spk_mix_dict)                # :578
        self._check_inference_only(*ctrls.values())
        B, F = f0_frames.shape[0], f0_frames.shape[1]
        u01 = torch.rand(B, F * self._hop, dtype=torch.float32, device=f0_frames.device)        # rand_like, :603
        signal, harmonic, noise = synth.sins_synth(
            f0_frames, st, ctrls["amplitudes"], ctrls["group_delay"], ctrls["noise_magnitude"], u01,
            self._sr, self._hop, noise_is_u01=True, want_components=self.return_components,
            fir_impl=self.fir_impl)
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
        st = synth.phase(f0_frames, self._sr, self._hop, initial_phase, infer)                 # :819-829
        ctrls, hidden = self.unit2ctrl(units_frames, f0_frames, st.phase_frames, volume_frames,
                                       spk_id=spk_id, spk_mix_dict=spk_mix_dict)                # :832
        self._check_inference_only(*ctrls.values())
        B, F = f0_frames.shape[0], f0_frames.shape[1]
        u01 = torch.rand(B, F * self._hop, dtype=torch.float32, device=f0_frames.device)        # rand_like, :854
        signal, harmonic, noise = synth.combsub_synth(
            f0_frames, st, ctrls["group_delay"], ctrls["harmonic_magnitude"], ctrls["noise_magnitude"], u01,
            self._sr, self._hop, noise_is_u01=True, want_components=self.return_components,
            fir_impl=self.fir_impl)
        return signal, hidden, (harmonic, noise)


def patch_reference():
    """Swap these classes (and the ddsp.core functions on the path) into an already-importable
    reference checkout so ``ddsp.vocoder.load_model``, ``main.py``, ``main_diff.py`` ... pick them up
    without edits.  Call before the reference scripts bind the names."""
    import ddsp.core as rcore
    import ddsp.vocoder as rvoc
    from . import core as hcore
    rvoc.Sins, rvoc.CombSub = Sins, CombSub
    for name in ("upsample", "remove_above_fmax", "frequency_filter", "fft_convolve",
                 "frequency_impulse_response"):
        setattr(rcore, "_reference_" + name, getattr(rcore, name))

        def dispatch(*a, __h=getattr(hcore, name), __r=getattr(rcore, name), **k):
            first = a[0] if a else next(iter(k.values()))
            return __h(*a, **k) if getattr(first, "is_cuda", False) else __r(*a, **k)
        setattr(rcore, name, dispatch)
    rvoc.upsample, rvoc.remove_above_fmax, rvoc.frequency_filter = (rcore.upsample, rcore.remove_above_fmax,
                                                                     rcore.frequency_filter)
    return rvoc
