"""Build recipe for libpocs.so (gfx950).  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libpocs.so"
SOURCES = ["pocs_kernels.hip", "pocs_host.hip"]
HEADERS = ["pocs_math.h", "pocs_model.h", "pocs_collide.h", "pocs_kernels.h", "pocs_command.hpp", "../../include/pocs.h"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libpocs.so cannot be built (there is no CPU fallback)")
    return exe


def stale():
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any((CSRC / f).stat().st_mtime > t for f in SOURCES + HEADERS)


def build_library(force=False, verbose=False):
    """Compile csrc/*.hip into libpocs.so next to this file; returns its path."""
    if not force and not stale():
        return LIB
    # One builder at a time (the ranks of a multi-GPU job all import the package at once), and the library
    # appears atomically: compile to a private name, then rename.
    import fcntl
    with open(PKG / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and not stale():                # another process built it while this one waited
            return LIB
        tmp = PKG / (".libpocs.%d.so" % os.getpid())
        cmd = [hipcc()] + FLAGS + [str(CSRC / s) for s in SOURCES] + ["-o", str(tmp)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if verbose or r.returncode:
            print(" ".join(cmd))
            print(r.stdout, r.stderr)
        if r.returncode:
            if tmp.exists():
                tmp.unlink()
            raise RuntimeError("hipcc failed building libpocs.so")
        os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
