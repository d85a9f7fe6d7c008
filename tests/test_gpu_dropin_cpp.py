"""C++ side of the drop-in: the reference's point2point / camera tests with the HIP-backed cost
classes under the unchanged LM loop (tests/cpp/dropin_point2point.cpp), run on the GPU box."""
import os
import subprocess

import numpy as np
import pytest

from tests import datasets as ds

pytestmark = pytest.mark.gpu

EXE = os.path.join(ds.ROOT, "tests", "cpp", "_build", "dropin_point2point")


def test_cpp_dropin_program(hip_lib, facade, tmp_path):
    assert os.path.exists(EXE), "build it with `make cpptests` (done by __graft_entry__.build())"
    src, _ = facade
    path = os.path.join(tmp_path, "facade.f64")
    np.ascontiguousarray(src, dtype="<f8").tofile(path)
    out = subprocess.run([EXE, path], capture_output=True, text=True, timeout=600)
    print(out.stdout[-4000:])
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert "SUMMARY failures=0" in out.stdout


def test_cpp_dropin_reference_model_tests(hip_lib):
    """The reference's curve-fitting / Powell / simple-model / loss / covariance / multi-objective /
    differentiation tests with the HIP cost classes (tests/cpp/dropin_models.cpp)."""
    exe = os.path.join(ds.ROOT, "tests", "cpp", "_build", "dropin_models")
    assert os.path.exists(exe), "build it with `make cpptests`"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout[-6000:])
    assert out.returncode == 0, out.stdout[-6000:] + out.stderr[-2000:]
    assert "failures=0" in out.stdout


def test_cpp_device_resident_optimizer(hip_lib):
    """The reference's known-answer LM tests with `hip::LevenbergMarquadtDevice` in place of
    `LevenbergMarquadtDynamic` (tests/cpp/dropin_device_lm.cpp): n = 2 / 4 / 6, float and double,
    one and two costs, against the known answers and against the host loop over the same costs."""
    exe = os.path.join(ds.ROOT, "tests", "cpp", "_build", "dropin_device_lm")
    assert os.path.exists(exe), "build it with `make cpptests`"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout[-8000:])
    assert out.returncode == 0, out.stdout[-8000:] + out.stderr[-2000:]
    assert ", 0 failures" in out.stdout


def test_cpp_binding_every_class_on_the_device(hip_lib):
    """tests/cpp/reference_headers_binding.cpp over host_api.hpp: every cost class and device model of
    the binding, float and double, built on the GPU and linearized with GemmanMCClure + a covariance
    set through the base class's setters.  (The CPU suite compiles the same source against the
    reference's own headers, where every construction must end in the NO_DEVICE exception.)"""
    exe = os.path.join(ds.ROOT, "tests", "cpp", "_build", "binding_all_classes")
    assert os.path.exists(exe), "build it with `make cpptests`"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout[-8000:])
    assert out.returncode == 0, out.stdout[-8000:] + out.stderr[-2000:]
    assert ", 0 failures; 28 costs built on a device, 0 refused" in out.stdout
