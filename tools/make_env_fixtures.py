#!/usr/bin/env python3
"""Extract the obstacle tables of the reference's two scenes (build container only: reads
/root/reference/pr2test2.env.xml and pr2custom.env.xml through envxml.load_env_xml) and commit them
as text fixtures -- data (numbers), not source -- under tests/golden/:

    tests/golden/pr2test2_env.txt    7 boxes: room walls, mid wall with the doorway, the small box
    tests/golden/pr2custom_env.txt   29 boxes: walls, 24 'spikes' turned +-60 degrees, one at 90

The GPU box has no /root/reference; tests/test_env_scenes.py loads these tables and runs them
through the HIP path against the oracle.  Usage: python tools/make_env_fixtures.py
"""
import sys
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")


def main():
    import pocs_amd  # noqa: F401  (registers the package alias)
    envxml = import_module("probability-of-collision-for-safe-planning_amd.envxml")
    out = ROOT / "tests" / "golden"
    for name in ("pr2test2", "pr2custom"):
        env = envxml.load_env_xml(REF / (name + ".env.xml"))
        envxml.write_env_txt(env, out / (name + "_env.txt"),
                             "%s.env.xml via envxml.load_env_xml, z-filter [0.05, 1.5]; skipped: %d external kinbodies"
                             % (name, len(env["skipped"])))
        print(name, env["boxes"].shape, "skipped", len(env["skipped"]))


if __name__ == "__main__":
    main()
