"""bench.py's CPU baseline leg, without a GPU: the reference's own compiled loop where oracle/_ref is built
(`kind: reference`, the restatement's figures under `port`), else the restatement (`kind: port`)."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_cpu_baseline_leg(pocs, plan, env, monkeypatch):
    sys.path.insert(0, str(ROOT))
    import bench
    monkeypatch.setenv("POCS_CPU_THREADS", "2")
    for path, K in (("gmm", 3), ("mc", 1)):
        c = bench.cpu_baseline(plan, env, K, 56, path, 1.5e5)
        assert c["unit"].startswith("particle-waypoint") and c["cores"] == 1 and c["value"] > 1e5 and "sample" in c
        if (ROOT / "oracle" / "_ref" / "libpocs_ref_loop.so").exists():
            assert c["kind"] == "reference" and "MCSimulator.h" in c["sample"] and "reference_error" not in c
            assert c["port"]["kind"] == "port" and c["port"]["value"] > 1e5
        else:
            assert c["kind"] == "port"
        assert c["all_cores"]["cores"] == 2 and c["all_cores"]["kind"] == "port" and c["all_cores"]["value"] > 0
