"""The in-register / LDS FFT plans of ddsp_svc_amd/csrc/fft_r.h (the building blocks of k_fir_blk, k_stft_filter, k_mel) on
their own, under the CPU emulator, against numpy.fft: the full three-exchange transform at 512, 1024, 2048 and 4096 points, the
two-exchange pair that stays in the scrambled layout S (forward_s with and without the pruned first pass, transposed), and
the two lockstep forms, which must equal their separate transforms bit for bit (same arithmetic, shared barriers)."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def plans():
    from tests.hipemu import build as emu
    lib_emu = emu.build()                                   # the emulator runtime (hipemu.cpp) lives in the product's emulator build
    out = os.path.join(emu.OUT, "libfft_plans_test.so")
    src = os.path.join(HERE, "hipemu", "fft_plans_test.hip")
    deps = [src, os.path.join(emu.CSRC, "fft_r.h"), os.path.join(emu.CSRC, "fft_1024p.h"), os.path.join(emu.CSRC, "fft2048.h"), os.path.join(HERE, "hipemu", "hip", "hip_runtime.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        obj = os.path.join(emu.OUT, "fft_plans_test.o")
        subprocess.run([emu._clang(), "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-I", emu.HERE, "-I", emu.CSRC,
                        "-Wno-unknown-attributes", "-Wno-unused-function", "-c", src, "-o", obj], check=True)
        subprocess.run([emu._clang(), "-shared", "-fPIC", obj, os.path.join(emu.OUT, "hipemu.o"), "-o", out], check=True)
    L = ctypes.CDLL(out)
    L.emu_fft_plan.argtypes = [ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 4
    L.emu_fft_plan.restype = ctypes.c_int

    def run(R, mode, a, b=None):
        n = 512 * R
        a = np.ascontiguousarray(a, dtype=np.complex64)
        b = np.ascontiguousarray(b if b is not None else np.zeros(n), dtype=np.complex64)
        o1, o2 = np.zeros(n, np.complex64), np.zeros(n, np.complex64)
        assert L.emu_fft_plan(R, mode, a.ctypes.data, b.ctypes.data, o1.ctypes.data, o2.ctypes.data) == 0
        return o1, o2
    return run


def _rand(n, seed, half=False):
    rng = np.random.default_rng(seed)
    z = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    if half:
        z[n // 2:] = 0
    return z


def _close(got, ref):
    ref = ref.astype(np.complex128)
    return np.sqrt(np.mean(np.abs(got - ref) ** 2)) <= 3e-7 * np.sqrt(np.mean(np.abs(ref) ** 2))


@pytest.mark.parametrize("R", [1, 2, 4, 8])
def test_full_transform(plans, R):
    z = _rand(512 * R, R)
    assert _close(plans(R, 0, z)[0], np.fft.fft(z.astype(np.complex128)))


@pytest.mark.parametrize("R", [1, 2, 4, 8])
def test_lockstep_pair_and_zero_padded_first_pass(plans, R):
    """forward2 = two forward calls behind shared barriers, bit for bit; the pruned first pass (upper half of the input zero,
    never read) against numpy and, in lockstep, against itself."""
    n = 512 * R
    a, b = _rand(n, 30 + R), _rand(n, 40 + R)
    one_a, one_b = plans(R, 0, a)[0], plans(R, 0, b)[0]
    two_a, two_b = plans(R, 10, a, b)
    assert np.array_equal(one_a, two_a) and np.array_equal(one_b, two_b)
    ah, bh = _rand(n, 50 + R, half=True), _rand(n, 60 + R, half=True)
    pruned = plans(R, 12, ah)[0]
    assert _close(pruned, np.fft.fft(ah.astype(np.complex128)))
    pa, pb = plans(R, 11, ah, bh)
    assert np.array_equal(pa, pruned) and np.array_equal(pb, plans(R, 12, bh)[0])


def test_layout_s_pair(plans):
    z = _rand(1024, 10)
    zh = _rand(1024, 11, half=True)
    assert _close(plans(2, 1, z)[0], np.fft.fft(z.astype(np.complex128)))
    assert _close(plans(2, 2, zh)[0], np.fft.fft(zh.astype(np.complex128)))
    assert _close(plans(2, 3, z)[0], np.fft.fft(z.astype(np.complex128)))          # the transposed factorisation is the DFT itself
    # forward_s then transposed of the conjugate = N times the input, conjugated (the inverse-by-forward trick of the kernels)
    spec = plans(2, 1, z)[0]
    back = np.conj(plans(2, 3, np.conj(spec))[0]) / 1024
    assert _close(back, z)


def test_lockstep_forms_equal_their_parts(plans):
    a, b = _rand(1024, 20, half=True), _rand(1024, 21, half=True)
    one_a, one_b = plans(2, 2, a)[0], plans(2, 2, b)[0]
    two_a, two_b = plans(2, 4, a, b)
    assert np.array_equal(one_a, two_a) and np.array_equal(one_b, two_b)
    v = _rand(1024, 22)
    inv, fwd = plans(2, 5, v, b)
    assert np.array_equal(inv, plans(2, 3, v)[0]) and np.array_equal(fwd, one_b)
    # the three-buffer form (the tap transform one exchange behind the inverse): the same arithmetic again
    inv3, fwd3 = plans(2, 13, v, b)
    assert np.array_equal(inv3, inv) and np.array_equal(fwd3, one_b)


def test_sign_carrying_layout(plans):
    """layout S- (one v_fmac_f32 with a DPP operand per value in the lane-pair step; odd threads hold negated values): with the
    sign undone the transforms are those of the plain layout -- the lane-pair step computes a0 - a1 as -(a1 - a0), exactly"""
    a, b = _rand(1024, 30, half=True), _rand(1024, 31, half=True)
    v = _rand(1024, 32)
    assert np.array_equal(plans(2, 6, a)[0], plans(2, 2, a)[0])
    assert np.array_equal(plans(2, 7, v)[0], plans(2, 3, v)[0])
    two_a, two_b = plans(2, 8, a, b)
    assert np.array_equal(two_a, plans(2, 2, a)[0]) and np.array_equal(two_b, plans(2, 2, b)[0])
    inv, fwd = plans(2, 9, v, b)
    assert np.array_equal(inv, plans(2, 3, v)[0]) and np.array_equal(fwd, plans(2, 2, b)[0])
    inv, fwd = plans(2, 14, v, b)
    assert np.array_equal(inv, plans(2, 3, v)[0]) and np.array_equal(fwd, plans(2, 2, b)[0])


def test_padded_row_plan(plans):
    """fft_1024p.h (k_fir_blk6): the same transforms on padded exchange rows -- one base register per exchange instead of XOR
    swizzles -- bit for bit; and its one-base form of the mirrored read against parked(-k) for every thread and slot"""
    a, b = _rand(1024, 70, half=True), _rand(1024, 71, half=True)
    v = _rand(1024, 72)
    assert np.array_equal(plans(2, 15, a)[0], plans(2, 2, a)[0])
    inv, fwd = plans(2, 16, v, b)
    assert np.array_equal(inv, plans(2, 3, v)[0]) and np.array_equal(fwd, plans(2, 2, b)[0])
    assert float(np.abs(plans(2, 17, a)[0][:128]).sum()) == 0.0


def test_emulator_selftest():
    """the emulator's own kernels (LDS reversal across a barrier, a wave scan through shuffles, one f32 MFMA, work-items
    that leave before a barrier): tests/hipemu/selftest.hip"""
    from tests.hipemu import build as emu
    emu.build()
    out = os.path.join(emu.OUT, "libemu_selftest.so")
    src = os.path.join(HERE, "hipemu", "selftest.hip")
    obj = os.path.join(emu.OUT, "selftest.o")
    if not os.path.exists(out) or os.path.getmtime(src) > os.path.getmtime(out):
        subprocess.run([emu._clang(), "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-I", emu.HERE, "-I", emu.CSRC,
                        "-Wno-unknown-attributes", "-Wno-unused-function", "-c", src, "-o", obj], check=True)
        subprocess.run([emu._clang(), "-shared", "-fPIC", obj, os.path.join(emu.OUT, "hipemu.o"), "-o", out], check=True)
    L = ctypes.CDLL(out)
    L.emu_selftest.restype = ctypes.c_int
    assert L.emu_selftest() == 0
