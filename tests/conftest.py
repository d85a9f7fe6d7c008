import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_tree():
    """The tests exercise built artefacts (the HIP library, the oracle, the C++ programs), which are
    kept out of the git history: build whatever is missing once per session, as
    __graft_entry__.build() does.  hipcc cross-compiles for gfx950 without a GPU."""
    lib = os.path.join(ROOT, "moptimizer_0_amd", "lib", "libmoptimizer_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", ROOT, "-j4", "all"], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "liboracle.so")):
        subprocess.check_call(["make", "-C", ROOT, "oracle"], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "tests", "cpp", "_build", "dropin_models")):
        subprocess.check_call(["make", "-C", ROOT, "cpptests"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_binding
    return oracle_binding.load()


@pytest.fixture(scope="session")
def facade():
    """The reference's façade cloud (tst/data/fachada.txt) and its transformed copy, as
    tst/point2point.cpp:86-123 builds them."""
    from tests import datasets
    return datasets.facade_pair()


@pytest.fixture(scope="session")
def hip_lib():
    import moptimizer_0_amd as m
    m.capi.load()
    if m.capi.device_count() < 1:
        pytest.fail("no HIP device visible: the -m gpu tests must run on the GPU box")
    return m
