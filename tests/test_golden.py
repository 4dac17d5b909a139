"""Committed golden vectors (tests/golden/*.json, made by tools/make_golden.py from the oracle):
the oracle must keep reproducing them (CPU), and the HIP path must reproduce them (GPU)."""
import json
from pathlib import Path

import numpy as np
import pytest

HERE = Path(__file__).resolve().parent
GOLD = sorted((HERE / "golden").glob("*.json"))


def load(p, pocs, env):
    """-> document, case, plan, collision world of a golden file (see tools/make_golden.py)."""
    doc = json.loads(p.read_text())
    c = doc["case"]
    plan = pocs.load_plan()
    if c.get("sub"):
        plan = dict(traj=plan["traj"][:c["W"]], odom=plan["odom"][:c["W"] - 1])
    elif c["W"] != 56:
        plan = pocs.resample_plan(plan, c["W"])
    if "env" in c:                            # a committed scene table instead of the bundled pr2test2 one
        env = pocs.load_env(HERE / "golden" / (c["env"] + "_env.txt"))
        if "footprint" in c:
            env = dict(env, footprint=list(c["footprint"]))
    return doc, c, plan, env


def test_there_are_golden_files():
    assert len(GOLD) >= 7
    names = {p.stem for p in GOLD}
    assert {"cfg1_bundled_K3_N1000", "cfg5_mc_W500_N100000", "pr2custom_K3_N2000"} <= names


@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.stem)
def test_oracle_reproduces_golden(path, pocs, orc, env):
    doc, c, plan, env = load(path, pocs, env)
    cfg = orc.config(plan, env, K=c["K"])
    if "gmm_probability" in doc:
        g = orc.run_gmm(cfg, c["seed"], c["N"])
        assert g["prob"] == doc["gmm_probability"]
        assert [float(v) for v in g["probs"]] == doc["gmm_waypoint_probabilities"]
        assert g["moments"][:, :, :2].astype(int).tolist() == doc["gmm_counts"]
        assert np.allclose(g["moments"][-1], doc["gmm_moments_last"], rtol=1e-15, atol=0)
    if "mc_collided" in doc:
        n_mc, hits, parts = orc.run_mc(cfg, c["seed"], c["N"], want_particles=True)
        assert n_mc == doc["mc_collided"]
        assert np.bincount(hits, minlength=1).tolist() == doc["mc_hits_histogram"]
        assert [float(np.cumsum(parts[:, j])[-1]) for j in range(3)] == doc["mc_particles_checksum"]
    chain = orc.host_chain(cfg, c["seed"])
    assert [float(v) for v in chain["mu"][-1]] == doc["chain_mu_last"]
    assert [float(v) for v in chain["cov"][-1]] == doc["chain_cov_last"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=lambda p: p.stem)
def test_gpu_reproduces_golden(path, pocs, env):
    doc, c, plan, env = load(path, pocs, env)
    with pocs.Context(0) as ctx:
        ctx.configure(plan, env, K=c["K"], N=c["N"], seed=c["seed"])
        if "gmm_probability" in doc:
            p = ctx.run_gmm_estimation()
            probs = ctx.waypoint_probabilities()
            counts = np.array([ctx.moments(w, c["K"])[:, :2] for w in range(c["W"])]).astype(int)
            last = ctx.moments(c["W"] - 1, c["K"])
            assert counts.tolist() == doc["gmm_counts"]                    # integer work: bit exact
            assert [float(v) for v in probs] == doc["gmm_waypoint_probabilities"]
            assert abs(p - doc["gmm_probability"]) < 1e-12
            assert np.allclose(last, doc["gmm_moments_last"], rtol=1e-6, atol=1e-6)   # free running, see DESIGN 8
        ctx.set_seed(c["seed"])
        p_mc = ctx.run_simulation()
        chain = ctx.host_chain(8)
        parts, hits = ctx.particles(c["N"])
    assert p_mc == doc["mc_probability"]
    assert np.bincount(hits, minlength=1).tolist() == doc["mc_hits_histogram"]
    assert [float(np.cumsum(parts[:, j])[-1]) for j in range(3)] == doc["mc_particles_checksum"]   # final particles, bit for bit
    assert [float(v) for v in chain["mu"][-1]] == doc["chain_mu_last"]
    assert [float(v) for v in chain["cov"][-1]] == doc["chain_cov_last"]
