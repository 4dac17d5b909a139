"""The oracle's EKF arithmetic against the REFERENCE's own text: `make -C oracle ref_ekf` compiles the member
functions of MCSimulator.h:368-553 and :868-929 (and the helpers of :43-69) -- cut out of the header where it
lies, none of them touches an OpenRAVE symbol -- against the vendored Armadillo (oracle/ref_ekf_harness.cpp).
Function by function on random inputs and on the wrap edge cases, and as a whole chain on the bundled plan.
Skips where oracle/_ref is absent (the GPU box gets the prebuilt library; a checkout without /root/reference
has none)."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

LIB = Path(__file__).resolve().parents[1] / "oracle" / "_ref" / "libpocs_ref_ekf.so"
pytestmark = pytest.mark.skipif(not LIB.exists(), reason="oracle/_ref/libpocs_ref_ekf.so not built (no /root/reference)")

TWO_PI = 2 * 3.14159265358979323846


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@pytest.fixture(scope="module")
def ref(pocs):
    lib = C.CDLL(str(LIB))
    lib.refe_angle_wrap.restype = C.c_double
    lib.refe_angle_wrap.argtypes = [C.c_double]
    lib.refe_observation.restype = C.c_double
    lib.refe_sample_observation.restype = C.c_double
    d = pocs.DEFAULTS
    lm = np.asarray(d["landmarks"], np.float64)
    lib.refe_configure(_p(np.asarray(d["alphas"], np.float64)), C.c_double(d["Q"]), _p(np.ascontiguousarray(lm[0])),
                       _p(np.ascontiguousarray(lm[1])), C.c_int(lm.shape[1]))
    return lib


def v3(*a):
    return np.array(a, np.float64)


def close(a, b, tol=1e-15):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b))))


def poses(rng, n):
    for _ in range(n):
        yield v3(rng.uniform(-4, 4), rng.uniform(-2, 2), rng.uniform(-1, 7))


def controls(rng, n):
    for _ in range(n):
        yield v3(rng.uniform(0, TWO_PI), rng.uniform(0.0, 0.3), rng.uniform(0, TWO_PI))


def test_angle_wrap_and_motion_model(ref, orc):
    for a in [0.0, -0.0, -0.01, TWO_PI, TWO_PI + 1e-12, -TWO_PI, 3 * TWO_PI + 0.5, -7.0, 6.27, 1e-300, -1e-300]:
        assert ref.refe_angle_wrap(C.c_double(a)) == orc.wrap_angle(a), a          # 2 pi itself is kept; -0.01 -> 6.27...
    rng = np.random.default_rng(3)
    for x, u in zip(poses(rng, 300), controls(rng, 300)):
        out = np.zeros(3)
        ref.refe_prediction(_p(x), _p(u), _p(out))
        assert close(out, orc.prediction(x, u)), (x, u)          # libm cos/sin vs the oracle's kernels: <= 1e-15
        x2 = x + v3(rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-0.5, 0.5))
        ref.refe_inverse_odometry(_p(x), _p(x2), _p(out))
        assert close(out, orc.inverse_odometry(x, x2)), (x, x2)


def test_noise_matrix_and_jacobians(ref, orc, pocs):
    """generateM_EKF, generateG_EKF, generateV_EKF as written (V's row 3 = [1 0 1]) through EKFpredict."""
    rng = np.random.default_rng(4)
    al = np.asarray(pocs.DEFAULTS["alphas"], np.float64)
    for mu, u in zip(poses(rng, 300), controls(rng, 300)):
        M = np.zeros(9)
        ref.refe_generate_M(_p(u), _p(M))
        Md = orc.generate_M(al, u)
        assert np.array_equal(M.reshape(3, 3), np.diag(Md)), u                     # same products, same order: exact
        G, V = np.zeros(9), np.zeros(9)
        ref.refe_generate_G(_p(mu), _p(u), _p(G))
        ref.refe_generate_V(_p(mu), _p(u), _p(V))
        V = V.reshape(3, 3)
        assert V[2, 0] == 1.0 and V[2, 1] == 0.0 and V[2, 2] == 1.0                # not Thrun's V: copy as written
        A = rng.normal(size=(3, 3))
        S = A @ A.T * 1e-3 + np.eye(3) * 1e-4
        pm, pS = np.zeros(3), np.zeros(9)
        ref.refe_ekf_predict(_p(mu), _p(np.ascontiguousarray(S.ravel())), _p(u), _p(M), _p(pm), _p(pS))
        om, oS = orc.ekf_predict(mu, S, u, Md)
        assert close(pm, om) and close(pS.reshape(3, 3), oS, 1e-14), (mu, u)


def test_gain_and_applied_control(ref, orc):
    """generateL + the applied control of EKF_GaussProp (:714-726), with the deviations that make the wrapped rotations
    jump (a required -0.01 rad becomes 6.27) and the exact-zero deviation of step 0 (gain denominator 0.1)."""
    rng = np.random.default_rng(5)
    cases = []
    for nominal, u in zip(poses(rng, 300), controls(rng, 300)):
        goal = np.asarray(orc.prediction(nominal, u))
        est = nominal + v3(rng.normal(0, 0.02), rng.normal(0, 0.02), rng.normal(0, 0.02))
        cases.append((nominal, est, goal, u))
    n0, u0 = v3(-3.4, -1.4, -0.0), v3(0.4636476090008061, 0.1118033988749895, 5.8195376981787801)
    cases.append((n0, n0.copy(), np.asarray(orc.prediction(n0, u0)), u0))                 # xhat == 0: gain = ubar / 0.1, applied = u*
    cases.append((n0, n0 + v3(0.0, 1e-3, 0.02), np.asarray(orc.prediction(n0, u0)), u0))   # one zero deviation
    for nominal, est, goal, u in cases:
        L9, app = np.zeros(9), np.zeros(3)
        ref.refe_generate_L(_p(nominal), _p(est), _p(goal), _p(u), _p(L9))
        ref.refe_applied_control(_p(nominal), _p(est), _p(goal), _p(u), _p(app))
        want = orc.applied_control(nominal, est, goal, u)
        assert close(np.diag(L9.reshape(3, 3)), want["gain"], 1e-13), (nominal, est)
        assert close(app, want["applied"], 1e-14), (nominal, est)


def test_measurement_rows_and_update(ref, orc, pocs):
    """makeHRow, observation and the eight sequential scalar updates of EKFupdate (in place, no angle wrap)."""
    d = pocs.DEFAULTS
    lm = np.asarray(d["landmarks"], np.float64)
    rng = np.random.default_rng(6)
    for mu in poses(rng, 200):
        for lid in range(8):
            H = np.zeros(3)
            ref.refe_make_h_row(_p(mu), C.c_int(lid), _p(H))
            dx, dy = mu[0] - lm[0, lid], mu[1] - lm[1, lid]
            q = dx * dx + dy * dy
            assert close(H, [dx / np.sqrt(q), dy / np.sqrt(q), 0.0]) and H[2] == 0.0
            assert close(ref.refe_observation(_p(mu), C.c_int(lid)), np.sqrt(q))
        A = rng.normal(size=(3, 3))
        S = A @ A.T * 1e-3 + np.eye(3) * 1e-4
        z = np.array([np.hypot(mu[0] - lm[0, l], mu[1] - lm[1, l]) + rng.normal(0, 0.2) for l in range(8)])
        nm, nS = np.zeros(3), np.zeros(9)
        ref.refe_ekf_update(_p(mu), _p(np.ascontiguousarray(S.ravel())), _p(z), C.c_int(8), _p(nm), _p(nS))
        om, oS = orc.ekf_update(mu, S, z, lm[0], lm[1], d["Q"])
        assert close(nm, om, 1e-13) and close(nS.reshape(3, 3), oS, 1e-12), mu


def test_noise_is_mean_plus_z_times_sqrt_variance(ref, orc, pocs):
    """sampleOdometry's variances are built from the control it is GIVEN (the applied one, :754) in the draw order
    r1, tr, r2; sampleObservation adds N(0, Q) to the range: on the reference's own tape of normals."""
    al = np.asarray(pocs.DEFAULTS["alphas"], np.float64)
    rng = np.random.default_rng(7)
    for seed, (x, u) in enumerate(zip(poses(rng, 200), controls(rng, 200))):
        noisy, new, tape = np.zeros(3), np.zeros(3), np.zeros(3)
        ref.refe_sample_odometry(_p(x), _p(u), C.c_uint(seed + 1), _p(noisy), _p(new), _p(tape))
        r1, tr, r2 = u
        var = [al[0] * r1 * r1 + al[1] * tr * tr, al[2] * tr * tr + al[3] * (r1 * r1 + r2 * r2), al[0] * r2 * r2 + al[1] * tr * tr]
        assert close(noisy, [u[j] + tape[j] * np.sqrt(var[j]) for j in range(3)]), (x, u)
        assert close(new, orc.prediction(x, noisy))
        t1 = np.zeros(1)
        zv = ref.refe_sample_observation(_p(x), C.c_int(seed % 8), C.c_uint(seed + 7), _p(t1))
        lm = np.asarray(pocs.DEFAULTS["landmarks"], np.float64)
        assert close(zv, np.hypot(x[0] - lm[0, seed % 8], x[1] - lm[1, seed % 8]) + t1[0] * np.sqrt(pocs.DEFAULTS["Q"]))


def test_whole_chain_on_the_bundled_plan(ref, orc, plan, env, pocs):
    """The oracle's host chain on trajectory.dat / odometry.dat (55 steps, three seeds), re-done step by step with the
    reference's functions fed with the oracle's own noise: M on the nominal control, gain and applied control from
    the running estimate, EKFpredict, EKFupdate on the eight noisy ranges -- estimate and covariance after every
    step within 1e-12."""
    cfg = orc.config(plan, env, K=1)
    traj, odom = np.asarray(plan["traj"], np.float64), np.asarray(plan["odom"], np.float64)
    cov0 = np.asarray(pocs.DEFAULTS["cov0"], np.float64)
    for seed in (1, 2, 3):
        ch = orc.host_chain(cfg, seed)
        mu, S = traj[0].copy(), cov0.copy()
        for i in range(len(odom)):
            M, app, pm, pS, nm, nS = np.zeros(9), np.zeros(3), np.zeros(3), np.zeros(9), np.zeros(3), np.zeros(9)
            ref.refe_generate_M(_p(np.ascontiguousarray(odom[i])), _p(M))
            ref.refe_applied_control(_p(np.ascontiguousarray(traj[i])), _p(mu), _p(np.ascontiguousarray(traj[i + 1])), _p(np.ascontiguousarray(odom[i])), _p(app))
            assert close(app, ch["applied"][i], 1e-12), (seed, i)
            assert close(np.diag(M.reshape(3, 3)), ch["Mdiag"][i], 1e-15), (seed, i)
            ref.refe_ekf_predict(_p(mu), _p(np.ascontiguousarray(S.ravel())), _p(app), _p(M), _p(pm), _p(pS))
            ref.refe_ekf_update(_p(pm), _p(pS), _p(np.ascontiguousarray(ch["z"][i])), C.c_int(8), _p(nm), _p(nS))
            assert close(nm, ch["mu"][i], 1e-12) and close(nS, ch["cov"][i].ravel(), 1e-11), (seed, i)
            mu, S = nm.copy(), nS.reshape(3, 3).copy()
