"""The OpenRAVE plugin translation unit itself (plugin/mcsimplugin_pocs.cpp), compiled against the test double
in tests/openrave_shim (OpenRAVE and Boost are not in this image) and linked with libpocs.so:
  * CPU: it compiles and links; the command line a handler builds from a stream with NOTHING behind the
    command name -- what the reference's two estimator commands look like (mcsimplugin.cpp:66-81,
    MCSimulation.py:241,243) -- is intact, while round 2's idiom demonstrably loses it;
  * GPU: the three plugin entry points, the scene walk on the boxes of pr2test2.env.xml, and every command of the
    reference through InterfaceBase::SendCommand, with the oracle's answers."""
import subprocess
from pathlib import Path

import pytest

from conftest import SAN_FLAGS, SAN_SUFFIX

ROOT = Path(__file__).resolve().parents[1]
PKG = ROOT / "probability-of-collision-for-safe-planning_amd"
EXE = ROOT / "tests" / ("_plugin_demo" + SAN_SUFFIX)


@pytest.fixture(scope="module")
def demo(pocs):
    pocs.load_library()
    src = ROOT / "tests" / "plugin_demo.cpp"
    deps = [src, ROOT / "plugin" / "mcsimplugin_pocs.cpp", PKG / "csrc" / "mcmodule.hpp", PKG / "csrc" / "scene_boxes.hpp",
            ROOT / "include" / "pocs.h", ROOT / "tests" / "openrave_shim" / "openrave" / "plugin.h"]
    if not EXE.exists() or any(d.stat().st_mtime > EXE.stat().st_mtime for d in deps):
        subprocess.run(["g++", "-O1", "-std=c++17"] + SAN_FLAGS + [ "-Wall", "-I" + str(ROOT / "tests" / "openrave_shim"), str(src), "-o", str(EXE),
                        "-L" + str(PKG), "-lpocs", "-Wl,-rpath," + str(PKG), "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return EXE


def test_adapter_tu_compiles_and_keeps_an_argument_less_command(demo):
    out = subprocess.run([str(demo), "--cmdline-only"], capture_output=True, text=True, check=True).stdout.splitlines()
    assert out[0] == "LINE [runGMMEstimation ]" and out[1] == "LINE2 [setQ  0.04]"
    assert out[2] == "OLD_IDIOM_OK 0"          # `line << name << ' ' << sinput.rdbuf()` sets failbit on an exhausted stream


@pytest.mark.gpu
def test_adapter_runs_every_reference_command(demo, orc, plan, env):
    out = subprocess.run([str(demo), str(PKG / "data" / "pr2test2_plan.txt"), "3000", "3", "99"],
                         capture_output=True, text=True, check=True)
    res = dict(ln.split(None, 1) for ln in out.stdout.strip().splitlines())
    assert res["ADVERTISED"] == "MCModule" and res["MYCOMMAND"] == "output"
    assert res["COMMANDS"] == "18 of 18"
    cfg = orc.config(plan, env, K=3)
    # the scene walked from the (shim) environment is the table of data/pr2test2_env.txt: same answers
    assert float(res["GMM"]) == orc.run_gmm(cfg, 99, 3000)["prob"]
    assert float(res["GMM2"]) == orc.run_gmm(cfg, (99 + 0x9E3779B97F4A7C15) % 2**64, 3000)["prob"]
    assert float(res["MC"]) == orc.run_mc(cfg, 99, 3000)[0] / 3000
    assert res["BADCMD"] == "0" and res["UNKNOWN"] == "0"
    assert "non-box geometries" in out.stderr and "collision world = 7 boxes" in out.stderr
