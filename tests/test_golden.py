"""Committed golden vectors (tests/golden/*.json, made by tools/make_golden.py from the oracle):
the oracle must keep reproducing them (CPU), and the HIP path must reproduce them (GPU)."""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = sorted((Path(__file__).resolve().parent / "golden").glob("*.json"))


def load(p, pocs):
    doc = json.loads(p.read_text())
    c = doc["case"]
    plan = pocs.load_plan()
    if c["W"] != 56:
        plan = pocs.resample_plan(plan, c["W"])
    return doc, c, plan


def test_there_are_golden_files():
    assert len(GOLD) >= 4


@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.stem)
def test_oracle_reproduces_golden(path, pocs, orc, env):
    doc, c, plan = load(path, pocs)
    cfg = orc.config(plan, env, K=c["K"])
    g = orc.run_gmm(cfg, c["seed"], c["N"])
    assert g["prob"] == doc["gmm_probability"]
    assert [float(v) for v in g["probs"]] == doc["gmm_waypoint_probabilities"]
    assert g["moments"][:, :, :2].astype(int).tolist() == doc["gmm_counts"]
    assert np.allclose(g["moments"][-1], doc["gmm_moments_last"], rtol=1e-15, atol=0)
    n_mc, hits, _ = orc.run_mc(cfg, c["seed"], c["N"])
    assert n_mc == doc["mc_collided"]
    assert np.bincount(hits, minlength=1).tolist() == doc["mc_hits_histogram"]
    chain = orc.host_chain(cfg, c["seed"])
    assert [float(v) for v in chain["mu"][-1]] == doc["chain_mu_last"]
    assert [float(v) for v in chain["cov"][-1]] == doc["chain_cov_last"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.stem)
def test_gpu_reproduces_golden(path, pocs, env):
    doc, c, plan = load(path, pocs)
    with pocs.Context(0) as ctx:
        ctx.configure(plan, env, K=c["K"], N=c["N"], seed=c["seed"])
        p = ctx.run_gmm_estimation()
        probs = ctx.waypoint_probabilities()
        counts = np.array([ctx.moments(w, c["K"])[:, :2] for w in range(c["W"])]).astype(int)
        last = ctx.moments(c["W"] - 1, c["K"])
        chain = ctx.host_chain(8)
        ctx.set_seed(c["seed"])
        p_mc = ctx.run_simulation()
        _, hits = ctx.particles(c["N"])
    assert counts.tolist() == doc["gmm_counts"]                    # integer work: bit exact
    assert [float(v) for v in probs] == doc["gmm_waypoint_probabilities"]
    assert abs(p - doc["gmm_probability"]) < 1e-12
    assert np.allclose(last, doc["gmm_moments_last"], rtol=1e-6, atol=1e-6)   # free running, see DESIGN 8
    assert p_mc == doc["mc_probability"]
    assert np.bincount(hits, minlength=1).tolist() == doc["mc_hits_histogram"]
    assert [float(v) for v in chain["mu"][-1]] == doc["chain_mu_last"]
    assert [float(v) for v in chain["cov"][-1]] == doc["chain_cov_last"]
