"""bench.py prints ONE JSON line with the contract's fields (GPU)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def run_bench(*extra, env=None):
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "5", "--warmup", "2", "--samples", "20000",
                          "--cpu-evals", "2e5", *extra], capture_output=True, text=True, check=True, cwd=str(ROOT),
                         env=env)
    lines = out.stdout.splitlines()
    assert len(lines) == 1, out.stdout            # ONE line on stdout, nothing else (library banners go to stderr)
    return json.loads(lines[0])


def test_default_line_has_every_field():
    d = run_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert sum(d["config"]["calls"]) == 5
    # SURVEY 8d's protocol: the median of >= 10 repeats of the timed pass, the spread beside it
    t = d["timing"]
    assert t["repeats"] >= 10 and t["ms_per_step_min"] <= t["ms_per_step_median"] <= t["ms_per_step_max"]
    assert d["ms_per_step"] == t["ms_per_step_median"] and t["timed_region_s"] > 0
    assert d["numerics"].startswith("v9") and "numerics v9" in d["numerics"] and "Philox4x32-7" in d["config"]["random_stream"]
    r = d["roofline"]
    # the limiter names the bound; the fraction is still the north star's: algorithmic bytes against the HBM peak
    assert r["bound"] == "fp64-issue+power" and r["reported_against"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert r["algorithmic_bytes_per_launch"] == 26 * r["evals_per_launch"]
    # 100 000 evaluations per launch here: no committed counter record within 2 x of that -> no extrapolated traffic
    assert r["traffic"] is None and r["traffic_source"] is None
    assert r["limiter"]["kind"].startswith("FP64") and r["concurrent_launches"] == 1 and r["waypoint_us"] > 0
    assert d["single_call_evals_per_s"] > 0 and "single_call" in d["config"]["value_is"]
    c = d["cpu_baseline"]
    # the reference's own compiled loop where oracle/_ref travelled with the tree, else the restatement
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    if (ROOT / "oracle" / "_ref" / "libpocs_ref_loop.so").exists():
        assert c["kind"] == "reference" and c["port"]["kind"] == "port" and c["port"]["value"] > 0, c
    assert d["value"] > 0 and abs(d["value"] - 20000 * 56 * 5 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-9
    # one GPU's share of the host and, where the host has more, every core the process may use; sized to the GPU part
    assert c["all_cores"]["cores"] <= 16 and ("host_cores" not in c or c["host_cores"]["cores"] == c["nproc"] > c["all_cores"]["cores"])
    assert "budget" in c


def test_full_size_line_carries_the_counter_records():
    """At the workload's own size the line carries the committed PMC figures: HBM bytes per launch (close to the
    algorithmic 26 B/eval) and the vector instructions per evaluation with the issue fraction they imply."""
    recs = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    if not any(v.get("numerics") == "v9" and v.get("evals_per_launch") in (10_000_000, 20_000_000) for v in recs.values()):
        pytest.skip("no counter record of the current numerics at this launch size yet")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"],
                         capture_output=True, text=True, check=True, cwd=str(ROOT))
    d = json.loads(out.stdout.splitlines()[-1])
    r = d["roofline"]
    # 20 runs per call go out as two launches of 10 side by side (POCS_OPT_SUB_BATCHES, the default at this size): the
    # roofline figures are the chip's -- both launches' bytes over the waypoint's period
    assert r["concurrent_launches"] == 2 and r["evals_per_launch"] == 10_000_000 and r["evals_per_period"] == 20_000_000
    assert r["algorithmic_bytes_per_period"] == 26 * 20_000_000 and abs(r["achieved"] - 520.0e6 / (r["avg_kernel_us"] * 1e-6) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is not None and "measured at" in r["traffic_source"]
    assert 0.9 < r["traffic"] / r["algorithmic_bytes_per_period"] < 1.2
    lim = r["limiter"]
    assert 100 < lim["valu_instr_per_eval"] < 200 and 0.3 < lim["valu_issue_frac"] < 1.0 and lim["clock_MHz"] > 1000
    # the two floors that meet at the power cap: the arithmetic alone (the same launches without the sample stores) and
    # the stream alone at this GPU's fill rate -- both below the launch as it is, neither negligible
    assert 0.5 * r["avg_kernel_us"] < lim["kernel_us_without_sample_stores"] < r["avg_kernel_us"]
    assert 0.4 * r["avg_kernel_us"] < lim["stream_alone_us_at_fill_rate"] < r["avg_kernel_us"]


def test_strong_scaling_line():
    """--scaling strong: the workload's samples in total (here: all of them on the one GPU)."""
    d = run_bench("--scaling", "strong", "--no-cpu-baseline")
    assert d["scaling"] == "strong" and d["config"]["total_samples_per_run"] == d["config"]["samples_per_gpu"] == 20000


def test_mc_workload_line():
    d = run_bench("--workload", "mc", "--no-cpu-baseline")
    assert "MC" in d["metric"] and d["roofline"]["bytes_per_eval"] == 56 and "cpu_baseline" not in d
    # 64 roll-outs of 20 000 particles = 36 MB of state: resident in the Infinity Cache between launches, so the line
    # does not call its bytes per second an HBM fraction
    r = d["roofline"]
    assert r["resident"] == "infinity-cache" and r["bound"] == "infinity-cache" and r["frac"] is None and r["achieved"] > 0
    d = run_bench("--workload", "mc", "--no-cpu-baseline", "--samples", "1200000", "--batch", "8", "--steps", "8", "--warmup", "8")
    r = d["roofline"]                                  # 8 x 1.2e6 x 28 B = 269 MB: past the cache, the non-temporal kernel, an HBM fraction
    assert r["resident"] == "hbm" and r["bound"] == "hbm" and 0.2 < r["frac"] < 1.0


def test_sharded_path_on_one_rank_prints_one_line():
    """The N>1 code path (torch.distributed over RCCL, per-waypoint all-reduce, two engines)
    rehearsed with world size 1: same contract, and RCCL's banner must not reach stdout."""
    import os
    env = dict(os.environ, POCS_FORCE_SHARDED="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    d = run_bench("--no-cpu-baseline", env=env)
    assert d["n_gpus"] == 1 and d["value"] > 0 and sum(d["config"]["calls"]) == 5


def test_gpus_2_started_by_hand_brings_up_its_own_ranks():
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment: the parent starts the two ranks
    itself (here both on the one card, gloo as the host channel), relays rank 0's ONE line, and that line says
    n_gpus 2 -- never a 1-GPU line for a 2-GPU request.  Without enough GPUs and without the rehearsal switch it
    refuses."""
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = run_bench("--gpus", "2", "--no-cpu-baseline", env=dict(env, POCS_FORCE_DEVICE="0", POCS_DIST_BACKEND="gloo",
                                                                POCS_SKIP_SINGLE="1", POCS_NO_BOARD_PROBE="1"))
    assert d["n_gpus"] == 2 and d["config"]["total_samples_per_run"] == 40000 and "tail" in d["config"]["exchange"]
    assert "whole call" in d["config"]["exchange"]                # ... replayed from a graph by the library, as on one GPU
    assert len(d["roofline"]["ranks_kernel_us"]["all"]) == 2
    # a scaling run that explains itself: the probe's outcome and time, the closers' waits for the other rank's moments
    # (in-kernel clock), the ranks' kernel times and their spread -- and the strong-scaling workload riding along
    assert d["exchange_probe"]["ok"] is True and d["exchange_probe"]["seconds"] > 0
    xw = d["exchange_wait_us"]
    assert xw["available"] and 0 <= xw["min_us"] <= xw["median_us"]["median"] <= xw["max_us"] and len(xw["median_us"]["per_rank"]) == 2
    assert "ONE CARD" in xw["status"]                            # values of a rehearsal: unmeasured on hardware
    assert d["roofline"]["skew_us"] == d["roofline"]["ranks_kernel_us"]["max"] - d["roofline"]["ranks_kernel_us"]["min"]
    st = d["strong"]
    assert st["scaling"] == "strong" and "cfg3" in st["workload"] and st["value"] > 0 and st["ms_per_step"] > 0
    assert len(st["ranks_kernel_us"]["all"]) == 2 and "tail" in st["exchange"] and st["exchange_wait_us"]["available"]
    assert 0.0 < st["probability"] < 1.0
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                         capture_output=True, text=True, cwd=str(ROOT), env=env)
    assert out.returncode != 0 and out.stdout.strip() == "" and "refusing" in out.stderr


def test_two_ranks_on_one_card_take_the_onehop_path():
    """bench.py as the driver launches it for N = 2 (one process per rank, RANK / WORLD_SIZE from the
    environment), both ranks on the ONE card of this box with gloo as the host channel: the one-hop probe
    succeeds, the shards exchange their moments in the sampling launch's tail, rank 0 prints the one line,
    and the probability is what one process gets for the whole mixture (N = 2 x 20000)."""
    import os
    port = str(29600 + os.getpid() % 300)
    procs = []
    for rank in (0, 1):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK=str(rank), POCS_FORCE_DEVICE="0", POCS_DIST_BACKEND="gloo", POCS_SKIP_SINGLE="1",
                   POCS_NO_BOARD_PROBE="1")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                                       "--samples", "20000", "--no-cpu-baseline"], stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True, cwd=str(ROOT), env=env))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-2000:] for o in outs]
    lines = outs[0][0].splitlines()
    assert len(lines) == 1 and outs[1][0].strip() == "", (outs[0][0], outs[1][0])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "tail" in d["config"]["exchange"] and d["config"]["engines_in_flight"] == 1, d["config"]
    assert d["config"]["total_samples_per_run"] == 40000
    one = run_bench("--no-cpu-baseline", "--samples", "40000", "--steps", "4")       # later flags win: N = 40000, 4 steps
    assert d["config"]["probability"] == one["config"]["probability"]


def test_two_ranks_through_the_collective_path():
    """The same two ranks with POCS_ONEHOP=0: one all-reduce per waypoint from Python (gloo here, RCCL on a node) --
    the fallback bench.py takes when the one-hop probe fails.  Its launches and its collectives must be ordered on the
    engine's stream (they were not when an engine had no stream of its own): the probability is
    what one process gets for the whole mixture."""
    import os
    d = None
    for onehop in ("0",):
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
        d = run_bench("--gpus", "2", "--no-cpu-baseline", "--steps", "2", "--warmup", "1",
                      env=dict(env, POCS_FORCE_DEVICE="0", POCS_DIST_BACKEND="gloo", POCS_SKIP_SINGLE="1",
                               POCS_NO_BOARD_PROBE="1", POCS_ONEHOP=onehop))
        assert d["n_gpus"] == 2 and "all-reduce" in d["config"]["exchange"].lower() or "rccl" in d["config"]["exchange"].lower(), d["config"]["exchange"]
    # two engines take the calls in turn: engine 0 ran one warm-up run and then the run whose probability the line
    # carries -- its context's run 1, which is what one process reports after one warm-up run of one
    one = run_bench("--no-cpu-baseline", "--samples", "40000", "--steps", "1", "--warmup", "1")
    assert d["config"]["engines_in_flight"] == 2 and d["config"]["probability"] == one["config"]["probability"]


def test_two_ranks_mc_workload():
    """--workload mc over two ranks (no data-path collective; one all-reduce of the hit counts): the proportion is
    what one process reports for all the particles."""
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = run_bench("--gpus", "2", "--workload", "mc", "--no-cpu-baseline", "--steps", "2", "--warmup", "2",
                  env=dict(env, POCS_FORCE_DEVICE="0", POCS_DIST_BACKEND="gloo", POCS_SKIP_SINGLE="1", POCS_NO_BOARD_PROBE="1"))
    one = run_bench("--workload", "mc", "--no-cpu-baseline", "--samples", "40000", "--steps", "2", "--warmup", "2")
    assert d["n_gpus"] == 2 and "MC" in d["metric"] and d["config"]["probability"] == one["config"]["probability"]


def test_two_ranks_strong_scaling():
    """--scaling strong over two ranks: the workload's samples in total, half per rank -- the probability of the same
    total in one process."""
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    d = run_bench("--gpus", "2", "--scaling", "strong", "--no-cpu-baseline", "--steps", "4", "--warmup", "2",
                  env=dict(env, POCS_FORCE_DEVICE="0", POCS_DIST_BACKEND="gloo", POCS_SKIP_SINGLE="1", POCS_NO_BOARD_PROBE="1"))
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_samples_per_run"] == 20000 and d["config"]["samples_per_gpu"] == 10000
    one = run_bench("--no-cpu-baseline", "--steps", "4", "--warmup", "2")
    assert d["config"]["probability"] == one["config"]["probability"]
