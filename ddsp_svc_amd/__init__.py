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
import os as _os
import sys as _sys
import warnings as _warnings


def _hardware_queues():
    """The synthesiser tails run on TWO streams (the noise branch beside the harmonic chain), and HIP maps streams onto
    ``GPU_MAX_HW_QUEUES`` hardware queues round robin (default 4).  A process that has opened an RCCL communicator owns about
    seven more streams, so with four queues the second stream lands on the caller's queue and a B = 32 step takes 0.359 ms
    instead of 0.325 (EXPERIMENTS.md 5.3).  The variable is read when the HIP runtime initialises -- the first device call,
    not ``import torch`` -- so importing this package before that point is enough; a user's own setting wins.  When the
    runtime is already up with fewer than eight queues the package says so once (nothing can be changed any more)."""
    have = _os.environ.get("GPU_MAX_HW_QUEUES")
    torch = _sys.modules.get("torch")
    up = bool(torch is not None and hasattr(torch, "cuda") and torch.cuda.is_initialized())
    if have is None and not up:
        _os.environ["GPU_MAX_HW_QUEUES"] = "8"
        return "set"
    try:
        enough = have is not None and int(have) >= 8
    except ValueError:
        enough = False
    if not enough and up and _os.environ.get("DDSP_HIP_QUIET") is None:
        _warnings.warn("ddsp_svc_amd: the HIP runtime was initialised with GPU_MAX_HW_QUEUES=%s; in a process that also holds an "
                       "RCCL communicator the two-stream synthesiser tails then share one hardware queue (~10 %% slower steps). "
                       "Export GPU_MAX_HW_QUEUES=8, or import ddsp_svc_amd before the first CUDA call." % (have or "unset (4)"),
                       RuntimeWarning, stacklevel=3)
        return "late"
    return "kept"


hardware_queues = _hardware_queues()

from . import _ffi, build, core, loss, mel, nsf_source, synth  # noqa: E402,F401

__version__ = "0.1.0"
