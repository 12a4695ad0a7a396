"""CPU checks: the gfx950 library builds, loads, and exports every symbol include/ddsp_hip.h declares;
argument validation of the C ABI (no compute calls here -- no GPU)."""
import ctypes
import os
import re

import pytest


def test_library_builds_and_exports_every_declared_symbol():
    from ddsp_svc_amd import _ffi, build
    path = build.build()
    assert os.path.exists(path)
    lib = _ffi.bind(ctypes.CDLL(path))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "ddsp_hip.h")).read()
    declared = set(re.findall(r"\b(ddsp_hip_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_ffi.SIGNATURES), declared ^ set(_ffi.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ddsp_hip_version() == 165


def test_argument_errors_are_reported_without_a_gpu():
    from ddsp_svc_amd import _ffi
    lib = _ffi.lib()
    assert lib.ddsp_hip_upsample(None, 1, 0, 1, 512, None, None) == -1          # F <= 0
    assert lib.ddsp_hip_upsample(None, 0, 4, 1, 512, None, None) == 0           # empty batch is a no-op
    assert lib.ddsp_hip_fft_convolve(None, 0, None, None, None, None, 1, 4, 512, 511, 0, None) == -1  # odd N
    assert lib.ddsp_hip_ir_table_bytes(256) == (2 * 256 * 256 + 510) * 4
    assert lib.ddsp_hip_synth_workspace_bytes(1, 4, 512, 256) > 2 * 4 * 512 * 4 + 4 * 510 * 4
    assert b"workspace" in lib.ddsp_hip_error_string(-4)
    assert lib.ddsp_hip_stft_workspace_bytes(2, 4, 512) == 2 * 4 * 512 * 4
    assert lib.ddsp_hip_mel_frames(20 * 512, 2048, 512) == 20 and lib.ddsp_hip_mel_frames(100, 2048, 512) == 1
    assert lib.ddsp_hip_fast_source(None, 1, 0, 512, 44100.0, None, None, None, None) == -1   # F <= 0
    assert lib.ddsp_hip_stft_filter(None, None, 0, None, 1025, None, 1025, None, 1024, None, 0, 1.0, None, 2048,
                                    1, 1, 1, 4, 512, None, None) == -1                      # row stride < n bins
    # the loss from the waveforms: sizes, frame counts and workspaces without a launch
    assert lib.ddsp_hip_stft_loss_table_bytes(1153) == (2 * 1153 + 4096) * 8 and lib.ddsp_hip_stft_loss_table_bytes(512) == (1024 + 1024) * 8
    assert lib.ddsp_hip_stft_loss_table_bytes(2049) == 0 and lib.ddsp_hip_stft_loss_table_bytes(1) == 0
    assert lib.ddsp_hip_stft_loss_frames(1000, 100, 100) == 10 and lib.ddsp_hip_stft_loss_frames(1000, 100, 25) == 37
    assert lib.ddsp_hip_stft_loss_frames(99, 100, 100) == 0
    assert lib.ddsp_hip_stft_loss_backward_ws_bytes(2, 1000, 100, 100) == 0
    assert lib.ddsp_hip_stft_loss_backward_ws_bytes(2, 1000, 100, 25) == 2 * 37 * 100 * 4
    assert lib.ddsp_hip_stft_loss(None, None, 1, 1000, 1000, 100, 100, None, 1.0, 1e-7, 1.0, None, 0, None, None, None, None,
                                  None) == -1                                                 # null pointers
    assert lib.ddsp_hip_stft_loss_tables(4096, None, None) == -3                              # size the plans do not reach


def test_knobs_of_generations_that_are_not_in_the_build_are_refused():
    """the product library ships one generation per kernel: the knobs that select a superseded one (A/B builds only) are an error,
    not a silent no-op -- an A/B run must not measure the same kernel twice under two names"""
    import ctypes as C
    from ddsp_svc_amd import _ffi
    lib = _ffi.lib()
    assert lib.ddsp_hip_set_tuning(b"BLK_WPS", 2) != 0 and lib.ddsp_hip_set_tuning(b"BLK_PADLDS", 4096) != 0
    assert lib.ddsp_hip_set_tuning(b"SINS_V1", 2) != 0
    assert lib.ddsp_hip_set_tuning(b"BLK_WPS", 0) == 0 and lib.ddsp_hip_set_tuning(b"SINS_V1", 0) == 0
    assert lib.ddsp_hip_set_tuning(b"NO_SUCH_KNOB", 1) != 0
    # where a fused tail call leaves its intermediates: valid offsets for a 256-bin CombSub step, none for other bin counts
    off = (C.c_longlong * 6)()
    assert lib.ddsp_hip_tail_layout(1, 2, 8, 512, 256, 256, 256, 0, 0, C.addressof(off)) == 1
    assert list(off)[0] == 0 and all(o > 0 and o % 256 == 0 for o in list(off)[1:])
    assert lib.ddsp_hip_tail_layout(1, 2, 8, 512, 256, 512, 256, 0, 0, C.addressof(off)) == 0 and all(o == -1 for o in off)
    assert lib.ddsp_hip_tail_layout(0, 2, 8, 512, 40, 256, 256, 0, 0, C.addressof(off)) == 1 and off[1] == -1 and off[3] == -1
    assert lib.ddsp_hip_tail_layout(1, 2, 8, 256, 256, 256, 256, 0, 0, C.addressof(off)) == 0          # another hop
    assert lib.ddsp_hip_tail_layout(1, 0, 8, 512, 256, 256, 256, 0, 0, C.addressof(off)) < 0


def test_host_tensors_are_rejected():
    import torch
    from ddsp_svc_amd import core
    with pytest.raises(RuntimeError, match="no CPU path"):
        core.upsample(torch.zeros(1, 4, 1), 512)
