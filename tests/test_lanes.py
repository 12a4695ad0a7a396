"""Sub-batches and lanes (csrc/api.hip, round 5; opt-in through knob LANE_ROWS -- measured slower than one batch at every size, kept
for callers that must bound the scratch).  Utterances are independent (core.py:120-182 has no op across the batch), so
a large CombSub / Sins tail can be issued as sub-batches of ~LANE_ROWS frames alternating between two lanes (the caller's stream
pair and a pair the library owns), each lane re-using ONE workspace slot.  The samples must be those of the unsplit call BIT
FOR BIT -- every split (even, ragged, one utterance per sub-batch), every output (signal, harmonic, noise), supplied noise and
the in-kernel draw (whose counter carries the utterance number), with and without the second lane -- and the scratch of a
split call must be two slots of one sub-batch whatever B is."""
import numpy as np
import pytest
import torch

from oracle import ddsp_oracle as O
from tests.backends import BACKENDS, dev  # noqa: F401

SR, HOP = 44100, 512


def _t(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _combsub_inputs(B, F, device, seed, n=(256, 256, 256)):
    f0 = O.synth_f0(B, F, SR, HOP, seed=seed)
    cg, ch, cn = O.synth_controls(B, F, list(n), seed=seed + 1)
    u = np.random.default_rng(seed + 2).random((B, F * HOP), dtype=np.float32)
    return (f0, cg, ch, cn, u), tuple(_t(a, device) for a in (f0, cg, ch, cn, u))


def _combsub(tensors, noise_seed=None, initial_phase=None):
    from ddsp_svc_amd import synth
    f0, cg, ch, cn, u = tensors
    st = synth.phase(f0, SR, HOP, initial_phase=initial_phase)
    if noise_seed is not None:
        return synth.combsub_synth(f0, st, cg, ch, cn, None, SR, HOP, noise_seed=noise_seed, noise_offset=5)
    return synth.combsub_synth(f0, st, cg, ch, cn, u, SR, HOP, noise_is_u01=True)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,F,rows", [(4, 24, 48), (5, 17, 34), (3, 9, 9), (6, 12, 24)])
def test_combsub_split_same_bits(dev, B, F, rows, knobs):
    arrays, tensors = _combsub_inputs(B, F, dev, seed=300 + B)
    ip = torch.linspace(-1.0, 2.0, B, device=dev)
    knobs("LANE_ROWS", 1)                                # never split
    whole = _combsub(tensors, initial_phase=ip)
    knobs("LANE_ROWS", rows)
    split = _combsub(tensors, initial_phase=ip)
    for a, b in zip(whole, split):
        assert torch.equal(a, b)
    knobs("LANES", 1)                                    # the same sub-batches in sequence on one lane, one slot
    serial = _combsub(tensors, initial_phase=ip)
    for a, b in zip(whole, serial):
        assert torch.equal(a, b)
    # and they are the reference's numbers (the oracle on the last utterance: the one a wrong row offset would hit)
    f0, cg, ch, cn, u = arrays
    noise = (u * np.float32(2) - np.float32(1)).astype(np.float32)
    ref = O.combsub_dsp(f0[-1:], cg[-1:], ch[-1:], cn[-1:], noise[-1:], SR, HOP, initial_phase=ip[-1:].cpu().numpy())
    got = split[0][-1:].cpu().numpy()
    e = float(np.sqrt(np.mean((got - ref["signal"]) ** 2)))
    assert e <= 1e-5 * float(np.sqrt(np.mean(ref["signal"] ** 2))) and e <= 1e-4, e


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_combsub_split_in_kernel_noise(dev, knobs):
    """noise=None: the draw's counter holds the utterance number, so a sub-batch must draw ITS utterances' numbers"""
    _, tensors = _combsub_inputs(5, 12, dev, seed=77)
    knobs("LANE_ROWS", 1)
    whole = _combsub(tensors, noise_seed=1234)
    knobs("LANE_ROWS", 24)
    split = _combsub(tensors, noise_seed=1234)
    for a, b in zip(whole, split):
        assert torch.equal(a, b)
    assert not torch.equal(split[2][0], split[2][2])     # (different utterances do draw different numbers)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("n", [(256, 256, 256), (65, 129, 65)])
def test_sins_split_same_bits(dev, n, knobs):
    from ddsp_svc_amd import synth
    B, F, H = 5, 14, 96
    f0 = O.synth_f0(B, F, SR, HOP, seed=11)
    ca, cg, cn = O.synth_controls(B, F, [H, n[0], n[2]], seed=12)
    u = np.random.default_rng(13).random((B, F * HOP), dtype=np.float32)
    f0t, cat, cgt, cnt, ut = (_t(a, dev) for a in (f0, ca, cg, cn, u))

    def step():
        st = synth.phase(f0t, SR, HOP)
        return synth.sins_synth(f0t, st, cat, cgt, cnt, ut, SR, HOP, noise_is_u01=True)
    knobs("LANE_ROWS", 1)
    whole = step()
    knobs("LANE_ROWS", 28)
    split = step()
    for a, b in zip(whole, split):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_split_views_and_components_off(dev, knobs):
    """controls as torch.split views of one [B,F,768] tensor (row stride 768), signal only"""
    from ddsp_svc_amd import synth
    B, F = 4, 20
    f0 = _t(O.synth_f0(B, F, SR, HOP, seed=5), dev)
    ctrl = torch.randn(B, F, 768, generator=torch.Generator().manual_seed(3)).to(dev)
    cg, ch, cn = torch.split(ctrl, [256, 256, 256], dim=-1)
    u = torch.rand(B, F * HOP, generator=torch.Generator().manual_seed(4)).to(dev)

    def step():
        st = synth.phase(f0, SR, HOP)
        return synth.combsub_synth(f0, st, cg, ch, cn, u, SR, HOP, noise_is_u01=True, want_components=False)[0]
    knobs("LANE_ROWS", 1)
    whole = step()
    knobs("LANE_ROWS", 40)
    assert torch.equal(whole, step())


def test_workspace_is_two_slots_of_a_sub_batch(knobs):
    """B = 64 x 10 s takes the scratch of B = 32 (two slots of 16 utterances), not twice it"""
    import ctypes
    from ddsp_svc_amd import _ffi
    from tests.hipemu import build as emu_build
    lib = _ffi.bind(ctypes.CDLL(emu_build.build()))
    F = 862
    try:
        lib.ddsp_hip_set_tuning(b"LANE_ROWS", 1)
        unsplit = {B: lib.ddsp_hip_synth_workspace_bytes(B, F, HOP, 256) for B in (16, 32, 64)}
        lib.ddsp_hip_set_tuning(b"LANE_ROWS", 0)                         # the default: never split
        assert lib.ddsp_hip_synth_workspace_bytes(64, F, HOP, 256) == unsplit[64]
        lib.ddsp_hip_set_tuning(b"LANE_ROWS", 14336)
        ws = {B: lib.ddsp_hip_synth_workspace_bytes(B, F, HOP, 256) for B in (8, 16, 32, 48, 64, 256)}
    finally:
        lib.ddsp_hip_set_tuning(b"LANE_ROWS", 0)
    assert ws[8] < ws[16] == unsplit[16]                  # below two sub-batches: the unsplit layout
    assert ws[32] == 2 * unsplit[16] and ws[64] == ws[32] and ws[48] == ws[32] and ws[256] == ws[32]
    assert ws[64] < unsplit[64]


@pytest.mark.gpu
def test_full_size_split_same_bits_gpu(knobs):
    """BASELINE cfg 2 / cfg 4 shapes on the MI355X: lanes of ~16 utterances against the unsplit call, bit for bit, twice (a race between
    the lanes' workspace slots would show as a difference between repeats)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = torch.device("cuda:0")
    for B in (32, 64, 40):
        _, tensors = _combsub_inputs(B, 862, dev, seed=B)
        knobs("LANE_ROWS", 1)
        whole = _combsub(tensors)
        knobs("LANE_ROWS", 14336)
        for _ in range(3):
            split = _combsub(tensors)
            for a, b in zip(whole, split):
                assert torch.equal(a, b)
        torch.cuda.synchronize()
