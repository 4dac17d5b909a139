"""The oracle restatement against the reference's OWN code: oracle/_ref/libpocs_ref.so is the
reference's vendored Armadillo headers and GM_Model.h compiled where they lie (oracle/Makefile
target `ref`, harness oracle/ref_harness.cpp).  CPU only.  Skips when the prebuilt library is
absent (it is built wherever /root/reference is mounted -- the build container only: nothing derived from
the reference travels to the GPU box)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from conftest import dp

REF = Path(__file__).resolve().parents[1] / "oracle" / "_ref" / "libpocs_ref.so"
pytestmark = pytest.mark.skipif(not REF.exists(), reason="oracle/_ref not built (reference not mounted)")


@pytest.fixture(scope="module")
def ref():
    lib = C.CDLL(str(REF))
    lib.ref_final_combine.restype = C.c_double
    return lib


def test_mean_cov_match_armadillo(ref, orc):
    rng = np.random.default_rng(31)
    for n in (2, 3, 17, 1000, 20000):
        X = np.ascontiguousarray(rng.normal(size=(n, 3)) * [0.03, 0.05, 0.02] + [-3.4, -1.4, 6.2])
        m, c = np.zeros(3), np.zeros(9)
        ref.ref_mean_cov(dp(X), n, dp(m), dp(c))
        om, oc = orc.cov_mean(X)
        assert np.array_equal(om, m)                                   # same running sums, same order
        # Armadillo forms A^T A through its gemm kernels (another summation order): both are the
        # single-pass N-1 formula, equal to the cancellation level of that formula
        assert np.allclose(oc.ravel(), c, rtol=1e-7, atol=1e-11)
        assert np.allclose(c.reshape(3, 3), np.cov(X.T), rtol=1e-6, atol=1e-12)


def test_normalise_matches_armadillo(ref, orc):
    for counts in ([[10, 0, 5], [90, 100, 95]], [[3, 3], [0, 0]], [[0], [7]], [[1, 2, 3, 4], [4, 3, 2, 1]]):
        cn = np.ascontiguousarray(counts, dtype=np.float64)
        K = cn.shape[1]
        w = np.zeros(K)
        ref.ref_normalise_rows(dp(cn), K, dp(w))
        assert np.array_equal(w, orc.normalise_l1(cn[1]))


def test_ekf_predict_products_match_armadillo(ref, orc):
    """V M V^T and G S G^T + R as Armadillo evaluates the reference's expressions (:874, :878)."""
    rng = np.random.default_rng(32)
    for _ in range(100):
        th, r1, tr = rng.uniform(0, 6.28), rng.uniform(0, 6.28), rng.uniform(0.05, 0.3)
        s, c = orc.sincos(th + r1)
        G = np.array([[1, 0, -tr * s], [0, 1, tr * c], [0, 0, 1.0]])
        V = np.array([[-tr * s, c, 0], [tr * c, s, 0], [1, 0, 1.0]])
        Md = np.abs(rng.normal(size=3)) * 1e-5
        B = rng.normal(size=(3, 3)); S = (B @ B.T + 0.5 * np.eye(3)) * 1e-3
        R, P = np.zeros(9), np.zeros(9)
        ref.ref_vmvt(dp(np.ascontiguousarray(V)), dp(np.ascontiguousarray(np.diag(Md))), dp(R))
        ref.ref_gsgt_plus_r(dp(np.ascontiguousarray(G)), dp(np.ascontiguousarray(S)), dp(R), dp(P))
        _, oP = orc.ekf_predict(np.array([0.3, -0.2, th]), S, np.array([r1, tr, 0.1]), Md)
        assert np.allclose(oP.ravel(), P, rtol=1e-13, atol=1e-20)


def test_scalar_update_matches_armadillo(ref, orc):
    """One landmark of EKFupdate (:896-921) written with the reference's Armadillo expressions."""
    rng = np.random.default_rng(33)
    for _ in range(100):
        mu = np.array([rng.uniform(-3, 3), rng.uniform(-1.5, 1.5), rng.uniform(0, 6.28)])
        B = rng.normal(size=(3, 3)); P = (B @ B.T + 0.5 * np.eye(3)) * 1e-3
        lx, ly = np.array([3.0]), np.array([-2.0])
        q = np.hypot(mu[0] - lx[0], mu[1] - ly[0])
        z = np.array([q + rng.normal(0, 0.2)])
        H = np.array([-(lx[0] - mu[0]) / q, -(ly[0] - mu[1]) / q, 0.0])
        m2, P2 = mu.copy(), np.ascontiguousarray(P).ravel().copy()
        ref.ref_scalar_update(dp(H), C.c_double(0.04), C.c_double(z[0] - q), dp(m2), dp(P2))
        om, oP = orc.ekf_update(mu, P, z, lx, ly, 0.04)
        assert np.allclose(om, m2, rtol=1e-13, atol=1e-16)
        assert np.allclose(oP.ravel(), P2, rtol=1e-12, atol=1e-19)


def test_final_combine_matches_armadillo(ref):
    rng = np.random.default_rng(34)
    p = np.ascontiguousarray(rng.random(56) * 0.05)
    want = ref.ref_final_combine(dp(p), 56)
    prod = 1.0
    for v in p:
        prod *= (1.0 - v)
    assert abs((1.0 - prod) - want) < 1e-15


def test_component_split_has_the_same_law_as_gm_model(ref, orc, plan, env, capfd):
    """GM_Model draws N component indices from std::discrete_distribution (GM_Model.h:89-93) and counts
    them; the build draws the counts directly (Multinomial(N, w) as conditional binomials, numerics
    v6).  Same law: compare the counts of both with the expectation in units of the binomial standard
    deviation."""
    cfg = orc.config(plan, env, K=3)
    state = orc.gmm_advance(cfg, orc.gmm_initial_state(cfg), None)
    w = np.array([0.2, 0.5, 0.3])
    state[:, 12] = w
    N = 60000
    counts = (C.c_int * 3)()
    ref.ref_gm_model_counts(dp(np.ascontiguousarray(w)), 3, N, 12345, counts)
    capfd.readouterr()                                     # GM_Model prints its weights
    _, _, _, comp = orc.gmm_waypoint(cfg, 9, 0, state, 0, N, want_samples=True)
    ours = np.bincount(comp, minlength=3)
    sd = np.sqrt(N * w * (1 - w))
    assert sum(counts) == N and ours.sum() == N
    assert np.all(np.abs(np.array(list(counts)) - N * w) < 5 * sd)
    assert np.all(np.abs(ours - N * w) < 5 * sd)
