"""Pin oracle/aten_chain.py (the torch-CPU walk of the reference's op chain that bench.py times as ``cpu_baseline`` kind
"aten-chain") to the reference-generated fixtures: same operators in the same order as the reference, so it reproduces
the reference's float32 outputs to rounding (and, on the torch build that made the fixtures, bit for bit)."""
import os

import numpy as np
import pytest
import torch

from oracle import aten_chain as A
from tests.test_baseline_shapes import cfg1_inputs, check_summary

SR, HOP = 44100, 512


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))


@pytest.mark.parametrize("name,infer", [("combsub_256.npz", True), ("combsub_128.npz", True), ("combsub_small_train.npz", False),
                                        ("sins_h128.npz", True), ("sins_h40_train.npz", False), ("combsub_mixed.npz", True)])
def test_tails_match_reference(golden_dir, name, infer):
    g = np.load(os.path.join(golden_dir, name))
    t = lambda k: torch.from_numpy(g[k])
    with torch.no_grad():
        if name.startswith("sins"):
            out = A.sins_tail(t("f0_frames"), t("ctrl_amplitudes"), t("ctrl_group_delay"), t("ctrl_noise_magnitude"),
                              t("noise"), SR, HOP, infer)
        else:
            out = A.combsub_tail(t("f0_frames"), t("ctrl_group_delay"), t("ctrl_harmonic_magnitude"),
                                 t("ctrl_noise_magnitude"), t("noise"), SR, HOP, infer)
    for got, key in zip(out, ("signal", "harmonic", "noise_out")):
        assert rms(got.numpy() - g[key]) <= 1e-6 * rms(g[key]), key


@pytest.mark.parametrize("kind", ["combsub", "sins"])
def test_cfg1_matches_reference(golden_dir, kind):
    g = np.load(os.path.join(golden_dir, f"cfg1_{kind}_infer1.npz"))
    f0, c, noise = cfg1_inputs(g, kind)
    fn = A.sins_tail if kind == "sins" else A.combsub_tail
    with torch.no_grad():
        out = fn(torch.from_numpy(f0), *(torch.from_numpy(a) for a in c), torch.from_numpy(noise), SR, HOP, True)
    for got, key in zip(out, ("signal", "harmonic", "noise_out")):
        check_summary(got.numpy(), g, key, 1e-6)


@pytest.mark.parametrize("n_mag", [65, 256])
def test_filters_match_reference(golden_dir, n_mag):
    g = np.load(os.path.join(golden_dir, f"filter_n{n_mag}.npz"))
    audio = torch.from_numpy(g["audio"])
    ap = torch.complex(torch.from_numpy(g["resp_re"]), torch.from_numpy(g["resp_im"]))
    mag = torch.from_numpy(g["mag"])
    zm = torch.complex(mag, torch.zeros_like(mag))
    hw = torch.from_numpy(g["half_width"]).unsqueeze(-1)
    for key, y in (("y_roll", A.filter_with_response(audio, ap, window=False)),
                   ("y_hann", A.filter_with_response(audio, zm)),
                   ("y_dyn", A.filter_with_response(audio, zm, True, hw))):
        assert rms(y.numpy() - g[key]) <= 1e-6 * rms(g[key]), key
