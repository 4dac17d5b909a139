import ctypes as C
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


# `make -C tests sanitize`: the host-side helpers are built with AddressSanitizer + UBSan under other names
SANITIZE = __import__("os").environ.get("POCS_SANITIZE") == "1"
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-g"] if SANITIZE else []
SAN_SUFFIX = "_san" if SANITIZE else ""


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
    out = Path(__file__).with_name("_host_harness%s.so" % SAN_SUFFIX)
    csrc = ROOT / "probability-of-collision-for-safe-planning_amd" / "csrc"
    deps = [src] + [csrc / n for n in ("pocs_math.h", "pocs_model.h", "pocs_collide.h")]
    if not out.exists() or any(d.stat().st_mtime > out.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O1" if SANITIZE else "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-mfma"] + SAN_FLAGS +
                       [str(src), "-o", str(out)], check=True)
    lib = C.CDLL(str(out))
    lib.hh_log.restype = C.c_double
    lib.hh_log.argtypes = [C.c_double]
    lib.hh_wrap.restype = C.c_double
    lib.hh_wrap.argtypes = [C.c_double]
    return lib


def dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))
