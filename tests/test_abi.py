"""The C-ABI library loads and exports every symbol include/pocs.h declares (CPU only: no
compute call is made here)."""
import ctypes as C
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def declared_functions():
    text = (ROOT / "include" / "pocs.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pocs_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_reference_command_surface():
    names = declared_functions()
    for twin in ("set_alphas", "set_q", "set_num_landmarks", "set_landmarks", "set_num_particles",
                 "set_initial_covariance", "set_path_length", "set_trajectory", "set_odometry",
                 "set_num_gaussians", "set_num_gmm_samples", "run_simulation", "run_gmm_estimation",
                 "send_command", "create", "destroy"):
        assert "pocs_" + twin in names


def test_library_exports_every_declared_symbol(pocs):
    lib = pocs.load_library()
    names = declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "libpocs.so does not export %s" % n
    assert set(pocs.SIGNATURES) == set(names), set(pocs.SIGNATURES) ^ set(names)
    assert b"gfx950" in lib.pocs_version()


def test_no_cpu_fallback_without_a_gpu(pocs):
    """On a box without a HIP device the product must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pocs.PocsError) as e:
        pocs.Context(0)
    assert e.value.code == -4 and "no CPU path" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """The oracle is the checker: nothing in the product may include, import, link or load it."""
    pkg = ROOT / "probability-of-collision-for-safe-planning_amd"
    bad = re.compile(r"#\s*include[^\n]*oracle|import\s+oracle|from\s+oracle|libpocs_oracle|CDLL\([^)]*oracle")
    files = [f for ext in ("*.py", "*.h", "*.hip", "*.cpp") for f in pkg.rglob(ext)] + [ROOT / "pocs_amd.py"]
    assert len(files) >= 8
    for f in files:
        assert not bad.search(f.read_text()), f
