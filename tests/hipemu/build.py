"""Compile the product's HIP sources as host C++ against the hipemu header (CPU emulation of the
device model, TEST INFRASTRUCTURE ONLY).  Output: tests/hipemu/_build/libddsp_hip_emu.so"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "ddsp_svc_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libddsp_hip_emu.so")
SOURCES = ["phase.hip", "exciter.hip", "ir.hip", "ir_pfa.hip", "ir_czt.hip", "fir.hip", "fir_fft.hip", "fir_blk.hip", "fir_blk_bwd.hip", "fir_bwd_direct.hip", "fir_fft_bwd.hip", "stft.hip", "mel.hip", "mel_czt.hip", "sinegen.hip", "loss.hip", "loss_czt.hip", "api.hip"]


def _clang():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", "/opt/rocm/llvm/bin/clang++"):
        if os.path.exists(c):
            return c
    raise RuntimeError("ROCm clang++ not found (needed for ext_vector_type in host code)")


def build(force=False, sanitize=False):
    """DDSP_EMU_EXTRA_FLAGS (e.g. ``-DDDSP_PFA_ROWS=14``): a variant of the sources, built beside the default one under its own name"""
    extra = os.environ.get("DDSP_EMU_EXTRA_FLAGS", "").split()
    out = OUT if not extra else OUT + "_" + "".join(c if c.isalnum() else "_" for c in "".join(extra))
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, os.path.basename(LIB))
    lib = lib.replace(".so", "_asan.so") if sanitize else lib
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "hip", "hip_runtime.h"),
            os.path.join(ROOT, "include", "ddsp_hip.h")]
    fresh = lambda: os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps)
    if not force and fresh():
        return lib
    # one builder at a time (pytest-xdist workers all arrive here after a source change): the others wait, then find it fresh
    import fcntl
    with open(os.path.join(out, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and fresh():
            return lib
        return _build_locked(out, lib, extra, sanitize)


def _build_locked(out, lib, extra, sanitize):
    flags = ["-x", "c++", "-std=c++17", "-O2", "-g", "-fPIC", "-ffp-contract=off", "-I", HERE, "-I", CSRC,
             "-Wno-unknown-attributes", "-Wno-unused-function", *extra]
    if sanitize:
        flags += ["-fsanitize=address", "-fno-omit-frame-pointer"]
    objs = []
    for s in SOURCES + ["../../tests/hipemu/hipemu.cpp"]:
        src = os.path.normpath(os.path.join(CSRC, s))
        o = os.path.join(out, os.path.basename(s).replace(".hip", "").replace(".cpp", "") + ("_asan.o" if sanitize else ".o"))
        subprocess.run([_clang(), *flags, "-c", src, "-o", o], check=True)
        objs.append(o)
    link = [_clang(), "-shared", "-fPIC", *objs, "-o", lib]
    if sanitize:
        link.insert(1, "-fsanitize=address")
    subprocess.run(link, check=True)
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, sanitize="--asan" in sys.argv))
