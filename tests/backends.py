"""Two ways to run the SAME host code + kernel sources in tests:

``gpu``  the product path: cuda tensors, hipcc-built libddsp_hip.so (tests marked ``gpu``).
``emu``  CPU tensors, the same .hip sources compiled as host C++ against tests/hipemu (kernel
         logic check only -- indexing, LDS staging, wave scans, MFMA fragment maps).  Installed by
         monkeypatching two module globals of ddsp_svc_amd._ffi; the product never does this.
"""
import ctypes

import pytest
import torch

BACKENDS = ["emu", pytest.param("gpu", marks=pytest.mark.gpu)]


@pytest.fixture
def dev(request, monkeypatch):
    backend = request.param
    import ddsp_svc_amd
    from ddsp_svc_amd import _ffi, core, synth
    if backend == "emu":
        from tests.hipemu import build as emu_build
        cdll = _ffi.bind(ctypes.CDLL(emu_build.build()))
        monkeypatch.setattr(_ffi, "_LIB", cdll)
        monkeypatch.setattr(_ffi, "check_device", lambda *t: None)
        monkeypatch.setattr(core, "_TABLES", {})
        # host tensors ARE the emulated device here: never hand them to a patched-in reference class
        from ddsp_svc_amd import vocoder
        monkeypatch.setattr(vocoder._SynthBase, "_to_reference", lambda self, f0: False)
        return torch.device("cpu")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
