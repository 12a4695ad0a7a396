import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def knobs():
    """Set launcher tuning knobs (``ddsp_hip_set_tuning``, include/ddsp_hip.h) for one test and restore the defaults
    afterwards.  ``knobs("BLK_RUN", 2)`` acts on whatever library ``_ffi.lib()`` returns at that moment (the emulator
    build under the ``emu`` backend, libddsp_hip.so on the GPU), so call it inside the test, after the ``dev`` fixture."""
    touched = []

    def set_knob(name, value):
        from ddsp_svc_amd import _ffi
        lib = _ffi.lib()
        assert lib.ddsp_hip_set_tuning(name.encode(), int(value)) == 0, name
        touched.append((lib, name))
    yield set_knob
    for lib, name in touched:
        lib.ddsp_hip_set_tuning(name.encode(), 0)
