import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def is_experimental_build():
    """True when the loaded libdissc_hip.so was built with DISSC_EXPERIMENTAL=1 (it then also carries the kernels whose gates
    failed: conv_s2tc.hip, respair_wino.hip's F(4,3) pair, the k = 3 instances of the F(2,3) / eight-point kernels, hipGraph replay)"""
    import ctypes
    from dissc_amd import lib
    v = ctypes.c_int(0)
    return lib.dissc_get_option(b"experimental", ctypes.byref(v)) == 0 and v.value == 1


@pytest.fixture
def experimental():
    if not is_experimental_build():
        pytest.skip("kernel of a failed gate: only in DISSC_EXPERIMENTAL=1 builds "
                    "(DISSC_EXPERIMENTAL=1 python -c 'import __graft_entry__ as g; g.build()')")
