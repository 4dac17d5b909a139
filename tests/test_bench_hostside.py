"""bench.py's host-side logic that needs no GPU: the parent of a hand-started multi-GPU job, and the sizing of the
CPU baseline."""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_gpus_n_refuses_where_the_node_has_fewer_gpus():
    """`python bench.py --gpus 2` started by hand on a node that shows fewer GPUs (this container: none) refuses --
    nothing on stdout, a reason on stderr, a non-zero exit -- instead of printing a smaller job's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "POCS_FORCE_DEVICE")}
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this node has two GPUs")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, cwd=str(ROOT), env=env, timeout=300)
    assert out.returncode != 0 and out.stdout.strip() == "" and "refusing" in out.stderr


def test_a_failing_rank_ends_the_job_at_once():
    """spawn_ranks polls its ranks: with every rank forced onto a device that does not exist here, the first one to fail
    ends the job -- no line on stdout, the failure named -- long before any collective's timeout."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    t0 = time.time()
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, cwd=str(ROOT), timeout=300,
                         env=dict(env, POCS_FORCE_DEVICE="0", POCS_DIST_BACKEND="gloo"))
    import torch
    if torch.cuda.is_available():
        assert out.returncode == 0 and json.loads(out.stdout.splitlines()[-1])["n_gpus"] == 2      # (a GPU box: the job simply runs)
    else:
        assert out.returncode == 1 and out.stdout.strip() == "" and "exited with" in out.stderr, out.stderr[-1500:]
        assert time.time() - t0 < 120


def test_cpu_baseline_respects_its_budget(monkeypatch):
    """The CPU baseline sized to a budget: a 3 s budget gives a sample that all legs together finish in about that
    (the GPU part of a run must not be a blip beside it), and a thrown reference run does not become the baseline."""
    sys.path.insert(0, str(ROOT))
    import importlib
    bench = importlib.import_module("bench")
    import pocs_amd
    monkeypatch.setenv("POCS_CPU_THREADS", "2")
    plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
    t0 = time.time()
    c = bench.cpu_baseline(plan, env, 3, 56, "gmm", 6.0e7, budget_s=3.0)
    dt = time.time() - t0
    assert dt < 12.0, dt                                   # (legs: reference, port, 2 threads, all cores; a loaded box stretches them)
    assert c["value"] > 0 and c["cores"] == 1 and "budget" in c and c["all_cores"]["cores"] == 2
    assert c["kind"] in ("reference", "port")
    if c["kind"] == "reference":
        assert 0.0 < c["reference_probability"] < 1.0 and c["port"]["value"] > 0
