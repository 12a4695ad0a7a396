"""ddsp_svc_amd -- MI355X-native DDSP harmonic-plus-noise synthesiser (drop-in for the
Sins/CombSub forward pass of yxlllc/DDSP-SVC).

  core      -- ddsp/core.py functions (upsample, frequency_filter, ...) on HIP kernels
  synth     -- phase state, exciters, fused Sins / CombSub DSP tails
  vocoder   -- nn.Module drop-ins + patch_reference()
  mel       -- nsf_hifigan.nvSTFT.STFT.get_mel (the cascade's waveform -> log-mel front-end)
  nsf_source -- nsf_hifigan.models.SourceModuleHnNSF (SineGen + merge), the vocoder's harmonic source
  loss      -- ddsp/loss.py SSSLoss / RSSLoss (the STFT of any size below 2049 as an in-kernel chirp-z transform, the loss and
               its gradient in the same kernels; torch.stft only above that)
  sharding  -- utterance sharding across the GPUs of a node (+ optional RCCL gather)
"""
from . import _ffi, build, core, loss, mel, nsf_source, synth  # noqa: F401

__version__ = "0.1.0"
