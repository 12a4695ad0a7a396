"""Streaming shapes (the real-time caller, gui.py:118-133: B = 1, a fraction of a second per call).  Below 4096 frames a
CombSub step is issued as THREE dependent launches instead of seven -- exciter and the three tap syntheses in one launch
(k_front_small, grid.y) | all-pass filter beside the noise filter (grid.y) | harmonic filter + noise -- and the phase state as
ONE launch instead of two.  Same kernels, same arguments: the results must be the SAME BITS as the batch layout's (knob SMALL_PATH = 1
switches the fused forms off), at F = 1, 3, 47, 200, from any host thread, and within the usual bars of the oracle."""
import threading

import numpy as np
import pytest
import torch

from oracle import ddsp_oracle as O
from tests.backends import BACKENDS, dev  # noqa: F401

SR, HOP = 44100, 512


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


def _inputs(B, F, device, seed):
    f0 = O.synth_f0(B, F, SR, HOP, seed=seed)
    f0[0] = np.clip(f0[0] * 2.1, 65, 800)
    cg, ch, cn = O.synth_controls(B, F, [256, 256, 256], seed=seed + 1)
    u = np.random.default_rng(seed + 2).random((B, F * HOP), dtype=np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return (f0, cg, ch, cn, u), tuple(t(a) for a in (f0, cg, ch, cn, u))


def _step(tensors):
    from ddsp_svc_amd import synth
    f0, cg, ch, cn, u = tensors
    st = synth.phase(f0, SR, HOP)
    sig, harm, nz = synth.combsub_synth(f0, st, cg, ch, cn, u, SR, HOP, noise_is_u01=True)
    return st, sig, harm, nz


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,F", [(1, 1), (1, 3), (1, 47), (1, 200), (3, 22), (2, 1000)])
def test_fused_launches_same_bits(dev, B, F, knobs):
    arrays, tensors = _inputs(B, F, dev, seed=100 + F)
    st, sig, harm, nz = _step(tensors)
    knobs("SMALL_PATH", 1)                               # the batch layout: one launch per kernel
    st2, sig2, harm2, nz2 = _step(tensors)
    assert torch.equal(st.phase0, st2.phase0) and torch.equal(st.phase_frames, st2.phase_frames)
    assert torch.equal(sig, sig2) and torch.equal(harm, harm2) and torch.equal(nz, nz2)
    # and they are the reference's numbers
    f0, cg, ch, cn, u = arrays
    noise = (u * np.float32(2) - np.float32(1)).astype(np.float32)
    ref = O.combsub_dsp(f0, cg, ch, cn, noise, SR, HOP)
    for got, key in ((sig, "signal"), (harm, "harmonic"), (nz, "noise")):
        e = rms(got.cpu().numpy() - ref[key])
        assert e <= 1e-5 * rms(ref[key]) and e <= 1e-4, (key, e, rms(ref[key]))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,F,H", [(1, 1, 40), (1, 47, 256), (2, 130, 128)])
def test_sins_fused_launches_same_bits(dev, B, F, H, knobs):
    """Sins at streaming shapes: sinusoid bank | both tap syntheses in one launch | noise filter | all-pass filter + noise"""
    from ddsp_svc_amd import synth
    f0 = O.synth_f0(B, F, SR, HOP, seed=9 + F)
    ca, cg, cn = O.synth_controls(B, F, [H, 256, 256], seed=F)
    u = np.random.default_rng(F).random((B, F * HOP), dtype=np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f0t, cat, cgt, cnt, ut = t(f0), t(ca), t(cg), t(cn), t(u)

    def step():
        st = synth.phase(f0t, SR, HOP)
        return synth.sins_synth(f0t, st, cat, cgt, cnt, ut, SR, HOP, noise_is_u01=True)
    a = step()
    knobs("SMALL_PATH", 1)
    b = step()
    assert all(torch.equal(x, y) for x, y in zip(a, b))
    ref = O.sins_dsp(f0, ca, cg, cn, (u * np.float32(2) - np.float32(1)).astype(np.float32), SR, HOP)
    e = rms(a[0].cpu().numpy() - ref["signal"])
    assert e <= 1e-5 * rms(ref["signal"]) and e <= 1e-4


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_small_path_is_taken_and_bounded(dev, knobs):
    """the fused forms stop at 4096 frames (B F): one frame below they run, at 4096 the batch layout does -- both equal to the
    batch layout bit for bit either way, so the switch-over cannot be seen in the numbers"""
    from ddsp_svc_amd import _ffi
    lib = _ffi.lib()
    # the workspace of a streaming shape holds the third tap buffer, a batch shape's does not
    small = lib.ddsp_hip_synth_workspace_bytes(1, 100, HOP, 256)
    assert small >= 4 * (3 * 100 * HOP + 3 * 100 * 510)
    big = lib.ddsp_hip_synth_workspace_bytes(8, 512, HOP, 256)
    assert big < 4 * (3 * 4096 * HOP + 3 * 4096 * 510) + 4096
    for B, F in ((1, 4095), (1, 4096)) if dev.type != "cpu" else ((1, 65), ):
        _, tensors = _inputs(B, F, dev, seed=7)
        _, sig, _, _ = _step(tensors)
        knobs("SMALL_PATH", 1)
        _, sig2, _, _ = _step(tensors)
        knobs("SMALL_PATH", 0)
        assert torch.equal(sig, sig2)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_streaming_session_same_bits(dev):
    """synth.StreamingCombSub -- every buffer allocated and every pointer bound once, a call = two C calls -- returns the
    functional API's numbers bit for bit, call after call, and rejects nothing the functional API accepts at that shape"""
    from ddsp_svc_amd import synth
    B, F = 1, 33
    sess = synth.StreamingCombSub(B, F, 256, 256, 256, SR, HOP, dev, want_components=True)
    for seed in (1, 2):
        _, tensors = _inputs(B, F, dev, seed=seed)
        f0, cg, ch, cn, u = tensors
        st, sig, harm, nz = _step(tensors)
        pst = sess.phase(f0)
        assert torch.equal(pst.phase0, st.phase0) and torch.equal(pst.phase_frames, st.phase_frames)
        s2, h2, n2 = sess.synth(f0, cg, ch, cn, u, noise_is_u01=True)
        assert torch.equal(s2, sig) and torch.equal(h2, harm) and torch.equal(n2, nz)
    # the session hands raw pointers to the C ABI: what does not fit its bound shape is refused, not read out of bounds
    packed = torch.cat([cg, ch, cn], -1)
    v = torch.split(packed, [256, 256, 256], -1)                          # views with frame stride 768: accepted
    s3, _, _ = sess.synth(f0, v[0], v[1], v[2], u, noise_is_u01=True)
    assert torch.equal(s3, sig)
    for bad in (lambda: sess.synth(f0, cg[:, :-1], ch, cn, u),            # a frame short
                lambda: sess.synth(f0, cg[..., :255], ch, cn, u),         # wrong bin count
                lambda: sess.synth(f0, cg.double(), ch, cn, u),           # wrong dtype
                lambda: sess.synth(f0, cg.transpose(1, 2).contiguous().transpose(1, 2), ch, cn, u),   # last dim not contiguous
                lambda: sess.synth(f0, cg, ch, cn, u[:, :-512]),          # noise of another length
                lambda: sess.phase(f0[:, :-1])):
        with pytest.raises(ValueError):
            bad()


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_from_another_host_thread(dev):
    """gui.py calls the synthesiser from the audio callback thread: the library keeps no per-thread state that a first call
    from a worker thread would miss (knobs are process-wide, events of the two-stream layout per thread and created on demand)"""
    _, tensors = _inputs(1, 40, dev, seed=3)
    _, want, _, _ = _step(tensors)
    got, err = [], []

    def work():
        try:
            for _ in range(3):
                got.append(_step(tensors)[1])
        except BaseException as e:                       # noqa: BLE001
            err.append(e)
    ts = [threading.Thread(target=work) for _ in range(2)]
    if dev.type == "cpu":                                # the CPU emulator's fibers are one-kernel-at-a-time: worker threads in turn
        for t in ts:
            t.start()
            t.join()
    else:                                                # on the MI355X: both threads at once
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    assert not err, err
    assert len(got) == 6 and all(torch.equal(g, want) for g in got)


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,F", [(1, 1), (1, 5), (1, 47), (2, 130)])
def test_superfast_streaming_layout_same_bits(dev, B, F, knobs):
    """CombSubSuperFast (configs/combsub.yaml as shipped; what gui.py runs) at streaming shapes: one frame pair per workgroup and the
    exciter made inside the filter's load path -- ONE launch behind fast_source instead of two; the same bits as the batch layout
    (knob SMALL_PATH = 1), and the oracle's numbers"""
    from ddsp_svc_amd import synth
    f0 = O.synth_f0(B, F, SR, HOP, seed=40 + F)
    f0[0] = np.clip(f0[0] * 1.7, 65, 800)
    hm, hp, nm, nph = O.synth_controls(B, F, [1025] * 4, seed=F + 1)
    gz = O.synth_gauss(B, F * HOP, seed=F + 2)
    w = torch.hann_window(2048)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = [t(a) for a in (f0, hm, hp, nm, nph, gz)]

    def step():
        fs = synth.fast_source(args[0], SR, HOP)
        return synth.combsubsuperfast_synth(args[0], fs, args[1], args[2], args[3], args[4], args[5], w.to(dev), SR, HOP)
    a = step()
    knobs("SMALL_PATH", 1)
    b = step()
    assert torch.equal(a, b)
    ref = O.combsubsuperfast_dsp(f0, hm, hp, nm, nph, gz, SR, HOP, 2048, w.numpy())["signal"]
    e = rms(a.cpu().numpy() - ref)
    assert e <= 1e-5 * rms(ref), (e, rms(ref))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
def test_superfast_streaming_session_same_bits(dev):
    """synth.StreamingCombSubSuperFast (the model gui.py runs): every buffer allocated and every pointer bound once, a call = two C
    calls -- the functional API's numbers bit for bit, call after call, split views accepted, what does not fit the bound shape refused"""
    from ddsp_svc_amd import synth
    B, F, n = 1, 47, 1025
    w = torch.hann_window(2048).to(dev)
    sess = synth.StreamingCombSubSuperFast(B, F, w, SR, HOP, dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for seed in (1, 2):
        f0 = t(O.synth_f0(B, F, SR, HOP, seed=60 + seed))
        hm, hp, nm, nph = (t(c) for c in O.synth_controls(B, F, [n] * 4, seed=seed))
        gz = t(O.synth_gauss(B, F * HOP, seed=seed + 2))
        fs = synth.fast_source(f0, SR, HOP)
        want = synth.combsubsuperfast_synth(f0, fs, hm, hp, nm, nph, gz, w, SR, HOP)
        st = sess.source(f0)
        assert torch.equal(st.rad_acc, fs.rad_acc) and torch.equal(st.phase_frames, fs.phase_frames)
        assert torch.equal(sess.synth(f0, hm, hp, nm, nph, gz), want)
    v = torch.split(torch.cat([hm, hp, nm, nph], -1), [n] * 4, -1)            # Unit2Control's split views: frame stride 4100
    assert torch.equal(sess.synth(f0, v[0], v[1], v[2], v[3], gz), want)
    for bad in (lambda: sess.synth(f0, hm[:, :-1], hp, nm, nph, gz), lambda: sess.synth(f0, hm[..., :1024], hp, nm, nph, gz),
                lambda: sess.synth(f0, hm.double(), hp, nm, nph, gz), lambda: sess.synth(f0, hm, hp, nm, nph, gz[:, :-512]),
                lambda: sess.source(f0[:, :-1])):
        with pytest.raises(ValueError):
            bad()


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,F", [(1, 33), (5, 820)])
def test_half_tap_rows_of_the_noise_filter_same_bits(dev, B, F, knobs):
    """The fused layouts keep the noise filter's tap rows -- a zero-phase response under the Hann window: even, tap N - j is tap j --
    as their first N/2 + 1 taps and the filter reads them mirrored; knob TAPS_FULL = 1 keeps whole rows.  Same bits, CombSub and
    Sins, at a streaming and at a batch shape (B F >= 4096: the paired filter launch)."""
    from ddsp_svc_amd import synth
    arrays, tensors = _inputs(B, F, dev, seed=7 + F)
    f0, cg, ch, cn, u = tensors
    amps = torch.from_numpy(O.synth_controls(B, F, [64], seed=F)[0]).to(dev)

    def both():
        st = synth.phase(f0, SR, HOP)
        c = synth.combsub_synth(f0, st, cg, ch, cn, u, SR, HOP, noise_is_u01=True)
        s = synth.sins_synth(f0, st, amps, cg, cn, u, SR, HOP, noise_is_u01=True)
        return list(c) + list(s)
    half = both()
    knobs("TAPS_FULL", 1)
    whole = both()
    assert len(half) == 6 and all(torch.equal(a, b) for a, b in zip(half, whole))


@pytest.mark.parametrize("dev", BACKENDS, indirect=True)
@pytest.mark.parametrize("B,F", [(1, 9), (2, 40), (3, 1400)])
def test_sins_filters_in_one_launch_same_bits(dev, B, F, knobs):
    """The Sins tail's two filters are ONE launch whose workgroups run the noise filter over their run of block pairs and then the
    all-pass filter over the same run, each thread reading back as addend the samples it has just stored (k_fir_blk6<.., SEQ>); knob
    SINS_SEQ = 1: the two launches it replaces.  Same bits in all three outputs -- one pair per workgroup, odd frame counts, and a
    batch shape whose runs have several pairs and a warm-up pass each."""
    from ddsp_svc_amd import synth
    arrays, tensors = _inputs(B, F, dev, seed=21 + F)
    f0, cg, ch, cn, u = tensors
    amps = torch.from_numpy(O.synth_controls(B, F, [64], seed=F + 1)[0]).to(dev)

    def run():
        st = synth.phase(f0, SR, HOP)
        return list(synth.sins_synth(f0, st, amps, cg, cn, u, SR, HOP, noise_is_u01=True))
    one = run()
    knobs("SINS_SEQ", 1)
    two = run()
    assert len(one) == 3 and all(torch.equal(a, b) for a, b in zip(one, two))
    assert float(one[0].abs().sum()) > 0 and torch.equal(one[0], two[0])
