"""The oracle's whole runs against the REFERENCE's own loop: `make -C oracle ref_loop` compiles MCSimulator.h's
estimator loop -- `EKF_GaussProp`, `truncateGMM`, the particle functions, the setters, the data members, the EKF
arithmetic -- from the header where it lies (oracle/ref_loop_harness.cpp lists the line ranges), against the
vendored Armadillo and GM_Model.h.  The one member that touches OpenRAVE, `checkCollision(const config&)`
(:269-285), is not taken; in its place the loop calls the oracle's 2-D predicate on this build's world (DESIGN.md 8,
row C1: the scene itself cannot be pinned).  So what is pinned here is everything AROUND the collision check: the
order of the loop (D1), which control feeds what, particles and their counters (P1-P3), the truncation and the
mixture bookkeeping (T1), the final product (F1) -- on the reference's own noise: the harness replays the
arma::randn calls of a run after the same seed and hands them over as tapes, and the oracle is run on those.
Skips where oracle/_ref is absent."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

LIB = Path(__file__).resolve().parents[1] / "oracle" / "_ref" / "libpocs_ref_loop.so"
pytestmark = pytest.mark.skipif(not LIB.exists(), reason="oracle/_ref/libpocs_ref_loop.so not built (no /root/reference)")

TWO_PI = 2 * 3.14159265358979323846


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def ref(orc, pocs, plan, env):
    import oracle
    return oracle.RefLoop(orc, pocs, plan, env)


@pytest.fixture(autouse=True)
def philox_again(orc):
    yield
    orc.set_tapes()


def angle_gap(a, b):
    d = np.abs(np.asarray(a) - np.asarray(b)) % TWO_PI
    return np.minimum(d, TWO_PI - d)


def test_mc_run_on_the_reference_tape(ref, orc):
    """runSimulation() with 300 particles, three seeds: the oracle on the same normals ends with the same belief
    (1e-10), the same particles (1e-9; headings modulo 2 pi), the same collision counter for every particle and the
    same proportion; and the reference checked N x W poses, as the oracle does."""
    cfg = ref.configure(particles=300, gaussians=1, samples=10)
    for seed in (11, 12, 13):
        r = ref.run_mc(seed)
        assert r["checked"] == 300 * ref.W
        orc.set_tapes(chain=r["chain"], init=r["init"])
        ch = orc.host_chain(cfg, 0)
        n, hits, parts = orc.run_mc(cfg, 0, 300, want_particles=True)
        assert np.allclose(ch["mu"][-1], r["mu"], rtol=0, atol=1e-10), (seed, ch["mu"][-1], r["mu"])
        assert np.allclose(ch["cov"][-1], r["cov"], rtol=0, atol=1e-10), seed
        assert np.allclose(parts[:, :2], r["particles"][:, :2], rtol=0, atol=1e-9), seed
        assert np.all(angle_gap(parts[:, 2], r["particles"][:, 2]) < 1e-9), seed
        assert np.array_equal(hits, r["hits"]), (seed, np.flatnonzero(hits != r["hits"]))
        assert n / 300 == r["p"]
        assert 0 < n < 300                                   # a run that separates colliding from free particles


def test_gmm_run_with_one_gaussian_on_the_reference_tape(ref, orc):
    """runGMMEstimation() with one Gaussian and 4000 samples (with one component every draw of the run is an
    arma::randn call of known shape, so the whole run can be replayed): belief, the truncated-and-propagated Gaussian
    after the last waypoint, every waypoint's probability as printed, and the final probability.  Sums are taken in
    different orders (Armadillo's mean / cov against the build's tree): 1e-9."""
    N = 4000
    cfg = ref.configure(particles=10, gaussians=1, samples=N)
    for seed in (21, 22):
        r = ref.run_gmm(seed, record=True)
        assert r["checked"] == N * ref.W
        orc.set_tapes(chain=r["chain"], gmm=r["gmm"], counts=r["counts"])
        ch = orc.host_chain(cfg, 0)
        o = orc.run_gmm(cfg, 0, N)
        assert np.allclose(ch["mu"][-1], r["mu"], rtol=0, atol=1e-10) and np.allclose(ch["cov"][-1], r["cov"], rtol=0, atol=1e-10)
        assert np.allclose(o["probs"], r["probs"], rtol=0, atol=5.1e-5), (seed, o["probs"], r["probs"])     # four printed decimals
        assert abs(o["prob"] - r["p"]) < 1e-12, (seed, o["prob"], r["p"])
        # the mixture after the last waypoint's truncation: the reference stores mean / cov of the free samples there
        # (:601-602); the oracle's last moments give the same through pocs_truncated_moments' formula
        m = o["moments"][-1][0]
        n = m[0]
        mean = m[2:5] / n
        assert np.allclose(mean, r["means"][0], rtol=0, atol=1e-9), (seed, mean, r["means"][0])
        cxx = (m[5] - m[2] * m[2] / n) / (n - 1)
        ctt = (m[10] - m[4] * m[4] / n) / (n - 1)
        assert abs(cxx - r["covs"][0][0]) < 1e-9 and abs(ctt - r["covs"][0][8]) < 1e-9
        assert 0.0 < r["p"] < 1.0


def test_gmm_run_with_three_gaussians_on_the_reference_tape(ref, orc):
    """Three Gaussians.  Which component a sample belongs to comes from GM_Model's own engine, but the run prints
    every waypoint's counts (GM_Model.h:95-96): with those the arma::randn calls of the run have known shapes and the
    whole run is replayed -- the oracle given the same counts and the same normals.  This is the mixture bookkeeping
    end to end against the reference's loop: per-component truncation, the weights from the survivor counts, the
    per-component EKF, the samples of a component as one block.  Per-waypoint probabilities as printed (four
    decimals), the final probability, the three Gaussians after the last truncation."""
    for N, seed, gen_seed in ((3000, 31, 5), (3000, 32, 6), (3000, 33, 7), (1000, 34, 8), (1000, 35, 9)):   # 1000: BASELINE configs[0]
        cfg = ref.configure(particles=10, gaussians=3, samples=N)
        r = ref.run_gmm(seed, gen_seed=gen_seed, record=True)
        assert r["checked"] == N * ref.W and r["counts"].shape == (ref.W, 3)
        assert r["counts"][0].min() > N / 4 and r["counts"][-1].min() >= 0             # equal weights at the start
        orc.set_tapes(chain=r["chain"], gmm=r["gmm"], counts=r["counts"])
        o = orc.run_gmm(cfg, 0, N)
        assert np.allclose(o["probs"], r["probs"], rtol=0, atol=5.1e-5), (seed, o["probs"], r["probs"])
        assert abs(o["prob"] - r["p"]) < 1e-12, (seed, o["prob"], r["p"])
        for k in range(3):
            m = o["moments"][-1][k]
            if m[0] >= 2:
                assert np.allclose(m[2:5] / m[0], r["means"][k], rtol=0, atol=1e-9), (seed, k)
                assert abs((m[8] - m[3] * m[3] / m[0]) / (m[0] - 1) - r["covs"][k][4]) < 1e-9, (seed, k)
        # the weights the oracle carries into each waypoint reproduce the counts' expectation: survivors / all survivors
        w_last = o["states"][-1][:, 12]
        free_prev = o["moments"][-2][:, 0]
        assert np.allclose(w_last, free_prev / free_prev.sum(), rtol=0, atol=1e-15)


def test_gmm_runs_with_three_gaussians_agree_in_law(ref, orc):
    """The oracle's OWN component counts (conditional binomials on Philox) against the reference's (N categorical
    draws on its engine), and everything downstream of them: 48 runs each way (2000 samples) on the same plan and
    world.  The final probability of a run depends mostly on the path the chain realises, so the two sets of runs
    are compared as samples of one distribution: means within four standard errors, and a two-sample
    Kolmogorov-Smirnov test."""
    from scipy import stats
    N, runs = 2000, 48
    cfg = ref.configure(particles=10, gaussians=3, samples=N)
    a = np.array([ref.run_gmm(100 + s, gen_seed=500 + s)["p"] for s in range(runs)])
    orc.set_tapes()
    b = np.array([orc.run_gmm(cfg, 9000 + s, N)["prob"] for s in range(runs)])
    se = np.sqrt(a.var(ddof=1) / runs + b.var(ddof=1) / runs)
    assert abs(a.mean() - b.mean()) < 4 * se, (a.mean(), b.mean(), se)
    assert stats.ks_2samp(a, b).pvalue > 1e-3, (np.sort(a), np.sort(b))


def test_loop_on_the_second_scene_with_a_small_footprint(orc, pocs, plan):
    """The same replay on the reference's other scene (pr2custom.env.xml: 29 boxes, 25 of them turned) with a
    footprint small enough to pass between them, two Gaussians, the first 21 waypoints: runGMMEstimation() and
    runSimulation() of the compiled loop against the oracle on the run's own noise."""
    import oracle
    from pathlib import Path as P
    env = pocs.load_env(P(__file__).resolve().parent / "golden" / "pr2custom_env.txt")
    env = dict(env, footprint=[0.0, 0.0, 0.12, 0.10])
    sub = dict(traj=plan["traj"][:21], odom=plan["odom"][:20])
    ref = oracle.RefLoop(orc, pocs, sub, env)
    N = 2500
    cfg = ref.configure(particles=200, gaussians=2, samples=N)
    r = ref.run_gmm(41, gen_seed=9, record=True)
    assert "error" not in r, r
    orc.set_tapes(chain=r["chain"], gmm=r["gmm"], counts=r["counts"])
    o = orc.run_gmm(cfg, 0, N)
    assert abs(o["prob"] - r["p"]) < 1e-12 and np.allclose(o["probs"], r["probs"], rtol=0, atol=5.1e-5)
    assert 0.0 < r["p"] < 1.0 and r["checked"] == N * 21
    m = ref.run_mc(42)
    orc.set_tapes(chain=m["chain"], init=m["init"])
    n, hits, parts = orc.run_mc(cfg, 0, 200, want_particles=True)
    assert np.array_equal(hits, m["hits"]) and n / 200 == m["p"] and 0 < n < 200
    assert np.allclose(parts[:, :2], m["particles"][:, :2], rtol=0, atol=1e-9)


def test_where_the_reference_has_no_defined_behaviour(orc, pocs, plan):
    """Recorded, not matched.  On the second scene the plan runs into the furniture from waypoint ~25 on: every
    sample of a Gaussian then collides, `noncollpoints` is empty, `mean` / `cov` of it are empty matrices
    (MCSimulator.h:592-598), and the next EKFpredict indexes into them: the reference's own loop ends in an Armadillo
    exception ("index out of bounds" / "incompatible matrix dimensions") -- inside OpenRAVE that is the end of the
    module.  This build defines the case (degenerate cases): a component with fewer than two survivors is
    retired with weight 0, and the estimation goes on; the oracle (and the HIP path, bit for bit) return a
    probability."""
    import oracle
    from pathlib import Path as P
    env = pocs.load_env(P(__file__).resolve().parent / "golden" / "pr2custom_env.txt")
    env = dict(env, footprint=[0.0, 0.0, 0.12, 0.10])
    sub = dict(traj=plan["traj"][:30], odom=plan["odom"][:29])
    ref = oracle.RefLoop(orc, pocs, sub, env)
    cfg = ref.configure(particles=10, gaussians=2, samples=2500)
    r = ref.run_gmm(41, gen_seed=9)
    assert r["p"] != r["p"] and ("out of bounds" in r["error"] or "incompatible" in r["error"]), r
    orc.set_tapes()
    o = orc.run_gmm(cfg, 41, 2500)
    assert 0.0 < o["prob"] <= 1.0
    assert np.any(o["states"][-1][:, 13] == 0.0)                 # a retired component is what kept it going


def test_replay_on_random_problems(orc, pocs, plan):
    """The replay as a fuzz: 60 random problems -- a stretch of the bundled plan (5 to 40 waypoints from a random
    start), its own noise levels, sensor variance, landmark set (2 to 8), initial covariance, one to three Gaussians,
    500 to 3000 samples, a random world of one to six boxes (a third of them turned) around the path and a random,
    mostly off-centre footprint -- each run by the reference's compiled loop (GMM and MC) and replayed by the oracle on
    the run's own noise.  Where the reference ends in an exception (a Gaussian without survivors, test above) the case
    counts as skipped; everywhere else: final probability 1e-12, printed per-waypoint probabilities, every particle's
    collision counter."""
    import oracle
    rng = np.random.default_rng(20260)
    done = skipped = informative = 0
    traj, odom = np.asarray(plan["traj"], np.float64), np.asarray(plan["odom"], np.float64)
    for case in range(60):
        W = int(rng.integers(5, 41))
        s = int(rng.integers(0, len(traj) - W + 1))
        sub = dict(traj=traj[s:s + W], odom=odom[s:s + W - 1])
        L = int(rng.integers(2, 9))
        params = dict(alphas=list(np.asarray(pocs.DEFAULTS["alphas"]) * 10.0 ** rng.uniform(-1, 1.5, 4)), Q=float(10.0 ** rng.uniform(-3, -0.5)),
                      landmarks=np.vstack([rng.uniform(-5, 5, L), rng.uniform(-5, 5, L)]),
                      cov0=np.diag(10.0 ** rng.uniform(-4, -2, 3)))
        M = int(rng.integers(1, 7))
        boxes = []
        for _ in range(M):
            i = int(rng.integers(0, W))
            ang, dist = rng.uniform(0, 2 * np.pi), rng.uniform(0.45, 1.3)         # beside the path, not on it
            off = dist * np.array([np.cos(ang), np.sin(ang)])
            boxes.append([sub["traj"][i][0] + off[0], sub["traj"][i][1] + off[1], rng.uniform(0.03, 0.3), rng.uniform(0.03, 0.3),
                          rng.choice([0.0, 0.0, rng.uniform(-3, 3)])])
        env = dict(footprint=[rng.choice([0.0, rng.uniform(-0.1, 0.1)]), rng.choice([0.0, rng.uniform(-0.1, 0.1)]),
                              rng.uniform(0.03, 0.18), rng.uniform(0.03, 0.18)], boxes=np.array(boxes))
        K, N = int(rng.integers(1, 4)), int(rng.integers(500, 3001))
        ref = oracle.RefLoop(orc, pocs, sub, env)
        cfg = ref.configure(particles=150, gaussians=K, samples=N, params=params)
        r = ref.run_gmm(1000 + case, gen_seed=77 + case, record=True)
        if "error" in r:
            skipped += 1
            continue
        orc.set_tapes(chain=r["chain"], gmm=r["gmm"], counts=r["counts"])
        o = orc.run_gmm(cfg, 0, N)
        if np.any(o["states"][:, :, 13] == 0.0):          # the oracle retired a component the reference limped on with (< 2 survivors)
            skipped += 1
            continue
        assert abs(o["prob"] - r["p"]) < 1e-12, (case, o["prob"], r["p"])
        assert np.allclose(o["probs"], r["probs"], rtol=0, atol=5.1e-5), case
        m = ref.run_mc(2000 + case)
        orc.set_tapes(chain=m["chain"], init=m["init"])
        n, hits, _ = orc.run_mc(cfg, 0, 150, want_particles=True)
        assert np.array_equal(hits, m["hits"]) and n / 150 == m["p"], case
        orc.set_tapes()
        done += 1
        informative += 0.0 < r["p"] < 1.0
    print('replayed', done, 'skipped', skipped, 'informative', informative)
    assert done >= 30 and informative >= 12, (done, skipped, informative)
