import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pocs():
    import pocs_amd
    return pocs_amd


@pytest.fixture(scope="session")
def orc():
    import oracle
    return oracle.Oracle()


@pytest.fixture(scope="session")
def plan(pocs):
    return pocs.load_plan()


@pytest.fixture(scope="session")
def env(pocs):
    return pocs.load_env()


@pytest.fixture(scope="session")
def hh():
    """Product inline functions compiled for the host (tests/host_harness.cpp)."""
    src = Path(__file__).with_name("host_harness.cpp")
    out = Path(__file__).with_name("_host_harness.so")
    csrc = ROOT / "probability-of-collision-for-safe-planning_amd" / "csrc"
    deps = [src] + [csrc / n for n in ("pocs_math.h", "pocs_model.h", "pocs_collide.h")]
    if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma",
                        str(src), "-o", str(out)], check=True)
    lib = C.CDLL(str(out))
    lib.hh_log.restype = C.c_double
    lib.hh_log.argtypes = [C.c_double]
    lib.hh_wrap.restype = C.c_double
    lib.hh_wrap.argtypes = [C.c_double]
    return lib


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))
