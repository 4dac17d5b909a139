"""Pin A1: the oracle's sampler transform against the reference's OWN mvnrnd / chol /
GM_Model::sampleNPoints.  oracle/_ref/libpocs_ref_lapack.so is the reference's vendored Armadillo
headers + GM_Model.h compiled where they lie, linked with the LAPACK/BLAS inside the image's scipy
wheel (oracle/Makefile target ref_lapack, harness oracle/ref_lapack_harness.cpp).  Armadillo's
generator is not ours, so each call hands back the tape of standard normals the reference consumed;
the restatement fed with that tape must land on the reference's points.  Build container only:
skips wherever oracle/_ref is absent (it does not travel to the GPU box)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from conftest import dp

REF = Path(__file__).resolve().parents[1] / "oracle" / "_ref" / "libpocs_ref_lapack.so"
pytestmark = pytest.mark.skipif(not REF.exists(), reason="oracle/_ref/libpocs_ref_lapack.so not built (reference not mounted)")
RTOL = 1e-15


def close_points(got, want, mean):
    """|difference| <= 1e-15 of the magnitudes involved (the reference rounds D*z, then adds M: two
    roundings where the fma chain has one per term, i.e. 1-2 ulp of max(|mean|, |point|))."""
    scale = np.maximum(np.abs(want), np.abs(np.asarray(mean))[None, :])
    return bool(np.all(np.abs(got - want) <= RTOL * scale))


@pytest.fixture(scope="module")
def refl():
    return C.CDLL(str(REF))


def spd(rng, scale):
    B = rng.normal(size=(3, 3))
    return (B @ B.T + 0.3 * np.eye(3)) * scale


def ref_mvnrnd(refl, mean, cov, n, seed):
    pts, tape = np.zeros((n, 3)), np.zeros((n, 3))
    ok = refl.refl_mvnrnd(dp(np.ascontiguousarray(mean)), dp(np.ascontiguousarray(cov).ravel()), n, C.c_uint(seed), dp(pts), dp(tape))
    return ok, pts, tape


def test_chol_lower_matches_lapack_potrf(refl, orc):
    rng = np.random.default_rng(41)
    for i in range(200):
        S = spd(rng, 10.0 ** rng.uniform(-6, 0))
        L9 = np.zeros(9)
        assert refl.refl_chol_lower(dp(np.ascontiguousarray(S).ravel()), dp(L9)) == 1
        ok, L = orc.chol3_lower(S)                            # l00 l10 l11 l20 l21 l22
        assert ok
        want = L9.reshape(3, 3)
        got = np.array([[L[0], 0, 0], [L[1], L[2], 0], [L[3], L[4], L[5]]])
        # potrf scales a column by a reciprocal where the restatement divides, and the entries below the
        # diagonal are differences: agreement to a few ulp of the factor's largest entry
        assert np.max(np.abs(got - want)) <= 4 * np.finfo(float).eps * np.max(np.abs(want)), i
        assert np.allclose(got @ got.T, S, rtol=1e-14, atol=1e-22)
    # the initial covariance of every run (0.001 I): exact agreement
    L9 = np.zeros(9)
    refl.refl_chol_lower(dp(np.ascontiguousarray(np.eye(3) * 0.001).ravel()), dp(L9))
    assert np.array_equal(L9.reshape(3, 3), np.diag([orc.chol3_lower(np.eye(3) * 0.001)[1][j] for j in (0, 2, 5)]))


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 50000])
def test_mvnrnd_on_the_reference_tape(refl, orc, n):
    """chol3_lower * z + mean (three fma chains) == arma::mvnrnd(M, C, n) on the same normals."""
    rng = np.random.default_rng(42 + n)
    for trial in range(5):
        mean = np.array([rng.uniform(-3.5, 3.5), rng.uniform(-1.5, 1.5), rng.uniform(0, 6.28)])
        cov = spd(rng, 10.0 ** rng.uniform(-5, -2))
        ok, pts, tape = ref_mvnrnd(refl, mean, cov, n, 1000 + trial)
        assert ok == 1
        got = orc.mvnrnd_tape(mean, cov, tape)
        assert got is not None
        assert close_points(got, pts, mean), (n, trial, np.max(np.abs(got - pts)))
    # a tape really is N(0, 1) draws in column-major order
    if n >= 50000:
        assert abs(tape.mean()) < 0.02 and abs(tape.std() - 1.0) < 0.02


def test_sample_n_points_on_the_reference_tape(refl, orc, capfd):
    """GM_Model::sampleNPoints (GM_Model.h:83-116): counts[k] points per component, one block after
    the other, each block = mvnrnd(mean_k, cov_k, counts[k]) drawing from Armadillo's generator in
    component order.  The restatement, block by block on the same tape, gives the same points."""
    rng = np.random.default_rng(43)
    K, N = 3, 4000
    means = np.array([[-3.4, -1.4, 0.0], [-3.3, -1.35, 0.02], [0.8, 1.05, 1.57]])
    covs = np.array([spd(rng, 1e-3) for _ in range(K)])
    w = np.array([0.2, 0.5, 0.3])
    counts = (C.c_int * K)()
    pts, tape = np.zeros((N, 3)), np.zeros((N, 3))
    assert refl.refl_sample_n_points(dp(np.ascontiguousarray(means)), dp(np.ascontiguousarray(covs)), dp(w), K, N,
                                     C.c_uint(7), C.c_uint(12345), counts, dp(pts), dp(tape)) == 1
    capfd.readouterr()
    counts = np.array(list(counts))
    assert counts.sum() == N
    sd = np.sqrt(N * w * (1 - w))
    assert np.all(np.abs(counts - N * w) < 5 * sd)                 # Multinomial(N, w), as the build draws it
    off = 0
    for k in range(K):
        got = orc.mvnrnd_tape(means[k], covs[k], tape[off:off + counts[k]])
        assert close_points(got, pts[off:off + counts[k]], means[k]), k
        off += counts[k]


def test_rank_deficient_covariance_reference_behaviour(refl, orc, capfd):
    """SURVEY 8a T1/A1 corner: a component left with 2 (3) survivors gets a truncated covariance of
    rank 1 (2) from cov().  What the reference's own mvnrnd then does, measured here on 300 random
    survivor sets per size: chol fails on most of them, and the eigen fallback
    (glue_mvnrnd_meat.hpp:100-132) REJECTS most of those too -- the single-pass covariance leaves a
    negative eigenvalue of order eps * |sum x^2| / n, below its tolerance -100 eps ||C||_F -- so
    mvnrnd returns false and GM_Model::sampleNPoints is left with an EMPTY matrix for that component
    (next stop: mean of a 3 x 0 matrix, then an out-of-bounds mu(2,0) in EKFpredict).  The fallback
    is therefore not a behaviour one can match; the build retires such a component on both sides
    (DESIGN.md 4 "degenerate cases").  With 4 survivors both always succeed."""
    rng = np.random.default_rng(1)
    for nsurv, lo, hi in ((2, 0.5, 1.0), (3, 0.2, 0.8), (4, 0.0, 0.0)):
        ref_fail = ours_fail = both = 0
        for _ in range(300):
            c = np.array([rng.uniform(-3.5, 3.5), rng.uniform(-1.5, 1.5), rng.uniform(0, 6.28)])
            rows = np.ascontiguousarray(c + rng.normal(size=(nsurv, 3)) * [0.03, 0.03, 0.02])
            mean, cov = orc.cov_mean(rows)
            ok, pts, _ = ref_mvnrnd(refl, mean, cov, 5, 5)
            ours = orc.chol3_lower(cov)[0]
            ref_fail += (ok == 0)
            ours_fail += (not ours)
            both += (ok == 0 and not ours)
            if ok == 0:
                assert not np.any(pts)                      # nothing was sampled
        capfd.readouterr()                                  # Armadillo's warnings
        assert lo * 300 <= ref_fail <= hi * 300, (nsurv, ref_fail)
        assert both == ref_fail                             # wherever the reference gives up, the build retires too
        assert (ours_fail == 0) == (nsurv == 4)
