"""ctypes loader for the CPU oracle (oracle/libpocs_oracle.so).  TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package.  See the header of pocs_oracle.c for what is restated and what is pinned.
"""
import ctypes as C
import os
import re
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SANITIZE = os.environ.get("POCS_SANITIZE") == "1"      # tests/Makefile `sanitize`: the ASan + UBSan build of the same file
LIB = HERE / ("libpocs_oracle_san.so" if SANITIZE else "libpocs_oracle.so")

MAX_K, MAX_L, NMOM, STATE = 8, 32, 11, 16
STREAM_CHAIN, STREAM_MCINIT, STREAM_GMM = 1, 2, 3


class OrcConfig(C.Structure):
    _fields_ = [("alphas", C.c_double * 4), ("Q", C.c_double), ("L", C.c_int), ("W", C.c_int),
                ("K", C.c_int), ("M", C.c_int), ("lx", C.c_double * MAX_L), ("ly", C.c_double * MAX_L),
                ("cov0", C.c_double * 9), ("fp", C.c_double * 4),
                ("traj", C.POINTER(C.c_double)), ("odom", C.POINTER(C.c_double)),
                ("boxes", C.POINTER(C.c_double))]


def build(force=False):
    src = HERE / "pocs_oracle.c"
    if force or not LIB.exists() or LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(HERE), LIB.name], check=True,
                       stdout=subprocess.DEVNULL)
    return LIB


_dp = C.POINTER(C.c_double)


def _p(a):
    return None if a is None else a.ctypes.data_as(_dp)


class Oracle:
    def __init__(self):
        build()
        L = self.lib = C.CDLL(str(LIB))
        L.orc_log.restype = C.c_double
        L.orc_log.argtypes = [C.c_double]
        L.orc_wrap_angle.restype = C.c_double
        L.orc_wrap_angle.argtypes = [C.c_double]
        L.orc_run_gmm.restype = C.c_double
        L.orc_run_mc.restype = C.c_longlong
        L.orc_collides.restype = C.c_int
        L.orc_radius2_unit32.restype = C.c_double
        L.orc_radius2_unit32.argtypes = [C.c_uint32]

    def set_sum_order(self, order):
        """1 (default): the build's summation tree (numerics v7; the HIP path agrees bit for bit);
        0: plain sequential sums over the free set in sample order."""
        self.lib.orc_set_sum_order(C.c_int(1 if order else 0))

    def set_tapes(self, chain=None, init=None, gmm=None, counts=None):
        """Test hook: standard normals from tapes instead of Philox (None = Philox).  chain: (W-1) x (3 + L), a
        step's r1 tr r2 z_0..; init: N x 3 (MC particles); gmm: W x N x 3 (a waypoint's samples, the components'
        blocks one after the other); counts: W x K samples per component, in place of the conditional binomials.
        The arrays are kept alive here until the next call."""
        keep = [None if a is None else np.ascontiguousarray(a, np.float64) for a in (chain, init, gmm)]
        cnt = None if counts is None else np.ascontiguousarray(counts, np.int64)
        self._tapes = keep + [cnt]
        stride = 0 if keep[0] is None else keep[0].shape[1]
        n_gmm = 0 if keep[2] is None else keep[2].shape[1]
        self.lib.orc_set_tapes(_p(keep[0]), C.c_int(stride), _p(keep[1]), _p(keep[2]), C.c_longlong(n_gmm),
                               None if cnt is None else cnt.ctypes.data_as(C.POINTER(C.c_longlong)))

    # ---- primitives -------------------------------------------------------------------
    def philox(self, ctr, key, rounds=10):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self.lib.orc_philox4x32(c, k, C.c_int(rounds), o)
        return list(o)

    def log(self, x):
        return self.lib.orc_log(C.c_double(x))

    def sincos(self, x):
        s, c = C.c_double(), C.c_double()
        self.lib.orc_sincos(C.c_double(x), C.byref(s), C.byref(c))
        return s.value, c.value

    def sincos_2pi_u32(self, w):
        s, c = C.c_double(), C.c_double()
        self.lib.orc_sincos_2pi_u32(C.c_uint32(w), C.byref(s), C.byref(c))
        return s.value, c.value

    def radius2_unit32(self, w):
        return self.lib.orc_radius2_unit32(C.c_uint32(w))

    def component_counts(self, K, state, seed, waypoint, n_total):
        state = np.ascontiguousarray(state, np.float64)
        out = np.zeros(K)
        self.lib.orc_component_counts(C.c_int(K), _p(state), C.c_uint64(seed), C.c_int(waypoint), C.c_longlong(n_total), _p(out))
        return out

    def binomial(self, n, p, seed, comp=0, waypoint=0):
        self.lib.orc_binomial.restype = C.c_double
        self.lib.orc_binomial.argtypes = [C.c_double, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32]
        return self.lib.orc_binomial(float(n), float(p), int(seed), int(comp), int(waypoint))

    def normal_pair_w2(self, wr, wa):
        a, b = C.c_double(), C.c_double()
        self.lib.orc_normal_pair_w2(C.c_uint32(wr), C.c_uint32(wa), C.byref(a), C.byref(b))
        return a.value, b.value

    def sincos_tab(self, x):
        s, c = C.c_double(), C.c_double()
        self.lib.orc_sincos_tab(C.c_double(x), C.byref(s), C.byref(c))
        return s.value, c.value

    def sincos_2pi_u32_tab(self, w):
        s, c = C.c_double(), C.c_double()
        self.lib.orc_sincos_2pi_u32_tab(C.c_uint32(w), C.byref(s), C.byref(c))
        return s.value, c.value

    def normal3(self, seed, index, waypoint, stream):
        z = (C.c_double * 3)()
        sp = C.c_uint32()
        self.lib.orc_normal3(C.c_uint64(seed), C.c_uint64(index), C.c_uint32(waypoint),
                             C.c_uint32(stream), z, C.byref(sp))
        return list(z), sp.value

    def sample_normals(self, seed, sample, waypoint, stream=STREAM_GMM):
        z = (C.c_double * 3)()
        sp = C.c_uint32()
        self.lib.orc_sample_normals(C.c_uint64(seed), C.c_uint64(sample), C.c_uint32(waypoint),
                                    C.c_uint32(stream), z, C.byref(sp))
        return list(z), sp.value

    def wrap_angle(self, a):
        return self.lib.orc_wrap_angle(C.c_double(a))

    def prediction(self, x, u):
        x = np.ascontiguousarray(x, np.float64); u = np.ascontiguousarray(u, np.float64)
        o = np.zeros(3)
        self.lib.orc_prediction(_p(x), _p(u), _p(o))
        return o

    def inverse_odometry(self, p1, p2):
        p1 = np.ascontiguousarray(p1, np.float64); p2 = np.ascontiguousarray(p2, np.float64)
        o = np.zeros(3)
        self.lib.orc_inverse_odometry(_p(p1), _p(p2), _p(o))
        return o

    def generate_M(self, alphas, u):
        al, u, o = (np.ascontiguousarray(a, np.float64) for a in (alphas, u, np.zeros(3)))
        self.lib.orc_generate_M(_p(al), _p(u), _p(o))
        return o

    def applied_control(self, nominal, estimated, goal, control):
        a, b, c, d = (np.ascontiguousarray(v, np.float64) for v in (nominal, estimated, goal, control))
        g, ap = np.zeros(3), np.zeros(3)
        self.lib.orc_applied_control(_p(a), _p(b), _p(c), _p(d), _p(g), _p(ap))
        return dict(gain=g, applied=ap)

    def ekf_predict(self, mu, S, u, Md):
        mu, S, u, Md = (np.ascontiguousarray(a, np.float64) for a in (mu, S, u, Md))
        pm, pS = np.zeros(3), np.zeros(9)
        self.lib.orc_ekf_predict(_p(mu), _p(S.ravel()), _p(u), _p(Md), _p(pm), _p(pS))
        return pm, pS.reshape(3, 3)

    def ekf_update(self, mu, S, z, lx, ly, Q):
        mu = np.array(mu, np.float64); S = np.array(S, np.float64).ravel().copy()
        z, lx, ly = (np.ascontiguousarray(a, np.float64) for a in (z, lx, ly))
        self.lib.orc_ekf_update(_p(mu), _p(S), _p(z), C.c_int(len(z)), _p(lx), _p(ly), C.c_double(Q))
        return mu, S.reshape(3, 3)

    def chol3_lower(self, S):
        S = np.ascontiguousarray(S, np.float64).ravel()
        L = np.zeros(6)
        ok = self.lib.orc_chol3_lower(_p(S), _p(L))
        return ok, L

    def mvnrnd_tape(self, mean, cov, z):
        """mvnrnd on a given tape of normals; z and the result are n x 3 (rows = columns of the 3 x n matrix)."""
        z = np.ascontiguousarray(z, dtype=np.float64)
        out = np.zeros_like(z)
        ok = self.lib.orc_mvnrnd_tape(_p(np.ascontiguousarray(mean, dtype=np.float64)),
                                      _p(np.ascontiguousarray(cov, dtype=np.float64).ravel()), _p(z),
                                      C.c_longlong(z.shape[0]), _p(out))
        return out if ok else None

    def cov_mean(self, rows):
        rows = np.ascontiguousarray(rows, np.float64)
        m, c = np.zeros(3), np.zeros(9)
        self.lib.orc_cov_mean(_p(rows), C.c_longlong(rows.shape[0]), _p(m), _p(c))
        return m, c.reshape(3, 3)

    def normalise_l1(self, v):
        v = np.ascontiguousarray(v, np.float64)
        o = np.zeros_like(v)
        self.lib.orc_normalise_l1(_p(v), C.c_int(len(v)), _p(o))
        return o

    def collides(self, x, y, th, fp, boxes):
        fp = np.ascontiguousarray(fp, np.float64)
        boxes = np.ascontiguousarray(boxes, np.float64).reshape(-1, 5)
        return bool(self.lib.orc_collides(C.c_double(x), C.c_double(y), C.c_double(th), _p(fp),
                                          _p(boxes), C.c_int(boxes.shape[0])))

    # ---- configuration ------------------------------------------------------------------
    def config(self, plan, env, K=3, alphas=None, Q=None, landmarks=None, cov0=None):
        """plan: dict(traj W x 3, odom (W-1) x 3); env: dict(footprint[4], boxes M x 5)."""
        from_defaults = DEFAULTS
        cfg = OrcConfig()
        alphas = from_defaults["alphas"] if alphas is None else alphas
        Q = from_defaults["Q"] if Q is None else Q
        lm = np.asarray(from_defaults["landmarks"] if landmarks is None else landmarks, np.float64)
        cov0 = np.asarray(from_defaults["cov0"] if cov0 is None else cov0, np.float64)
        for i in range(4):
            cfg.alphas[i] = alphas[i]
        cfg.Q = Q
        cfg.L = lm.shape[1]
        for i in range(cfg.L):
            cfg.lx[i] = lm[0, i]; cfg.ly[i] = lm[1, i]
        traj = np.ascontiguousarray(np.asarray(plan["traj"], np.float64).T)      # 3 x W by component
        odom = np.ascontiguousarray(np.asarray(plan["odom"], np.float64).T)
        boxes = np.ascontiguousarray(np.asarray(env["boxes"], np.float64).reshape(-1, 5))
        cfg.W = traj.shape[1]; cfg.K = K; cfg.M = boxes.shape[0]
        for i in range(9):
            cfg.cov0[i] = cov0.ravel()[i]
        for i in range(4):
            cfg.fp[i] = env["footprint"][i]
        cfg.traj = _p(traj); cfg.odom = _p(odom); cfg.boxes = _p(boxes)
        cfg._keep = (traj, odom, boxes)
        return cfg

    # ---- whole paths --------------------------------------------------------------------
    def host_chain(self, cfg, seed):
        n = max(cfg.W - 1, 1)
        out = dict(applied=np.zeros((n, 3)), Mdiag=np.zeros((n, 3)), noisy=np.zeros((n, 3)),
                   z=np.zeros((n, max(cfg.L, 1))), mu=np.zeros((n, 3)), cov=np.zeros((n, 9)))
        self.lib.orc_host_chain(C.byref(cfg), C.c_uint64(seed), _p(out["applied"]), _p(out["Mdiag"]),
                                _p(out["noisy"]), _p(out["z"]), _p(out["mu"]), _p(out["cov"]))
        if cfg.L and out["z"].shape[1] != cfg.L:
            out["z"] = out["z"]
        return out

    def run_mc(self, cfg, seed, N, first=0, count=None, want_particles=False):
        count = N if count is None else count
        hits = np.zeros(max(count, 1), np.uint32)
        parts = np.zeros((max(count, 1), 3)) if want_particles else None
        n = self.lib.orc_run_mc(C.byref(cfg), C.c_uint64(seed), C.c_longlong(first), C.c_longlong(count),
                                hits.ctypes.data_as(C.POINTER(C.c_uint32)), _p(parts))
        return n, hits[:count], (parts[:count] if parts is not None else None)

    def run_gmm(self, cfg, seed, N, want_samples=False):
        W, K = cfg.W, cfg.K
        probs = np.zeros(W); mom = np.zeros((W, K, NMOM)); states = np.zeros((W, K, STATE))
        samples = np.zeros((N, 3)) if want_samples else None
        flags = np.zeros(N, np.int16) if want_samples else None
        p = self.lib.orc_run_gmm(C.byref(cfg), C.c_uint64(seed), C.c_longlong(N), _p(probs), _p(mom),
                                 _p(states), _p(samples),
                                 None if flags is None else flags.ctypes.data_as(C.POINTER(C.c_int16)))
        return dict(prob=p, probs=probs, moments=mom, states=states, samples=samples, flags=flags)

    def gmm_initial_state(self, cfg):
        s = np.zeros((cfg.K, STATE))
        self.lib.orc_gmm_initial_state(C.byref(cfg), _p(s))
        return s

    def gmm_waypoint(self, cfg, seed, w, state, first, count, want_samples=False, n_total=None):
        """n_total: samples of the whole mixture (the component counts add up to it); default = the shard."""
        n_total = first + count if n_total is None else n_total
        state = np.ascontiguousarray(state, np.float64)
        mom = np.zeros((cfg.K, NMOM))
        samples = np.zeros((max(count, 1), 3)) if want_samples else None
        flags = np.zeros(max(count, 1), np.int16) if want_samples else None
        comp = np.zeros(max(count, 1), np.int8) if want_samples else None
        self.lib.orc_gmm_waypoint(C.byref(cfg), C.c_uint64(seed), C.c_int(w), _p(state),
                                  C.c_longlong(first), C.c_longlong(count), C.c_longlong(n_total), _p(mom), _p(samples),
                                  None if flags is None else flags.ctypes.data_as(C.POINTER(C.c_int16)),
                                  None if comp is None else comp.ctypes.data_as(C.POINTER(C.c_int8)))
        if want_samples:
            return mom, samples[:count], flags[:count], comp[:count]
        return mom

    def gmm_advance(self, cfg, prev, moments, u=None, Md=None, z=None):
        prev = np.ascontiguousarray(prev, np.float64)
        nxt = np.zeros_like(prev)
        moments = None if moments is None else np.ascontiguousarray(moments, np.float64)
        u, Md, z = (None if a is None else np.ascontiguousarray(a, np.float64) for a in (u, Md, z))
        self.lib.orc_gmm_advance(C.byref(cfg), _p(prev), _p(moments), _p(u), _p(Md), _p(z), _p(nxt))
        return nxt


# Parameters of every published run: gaussprop.py:36,39,45-46,56 and
# finalpaper/analysis/GMMsimReport_3Gaussians.txt:1-12.
DEFAULTS = dict(
    alphas=[0.00025 ** 2, 0.0025 ** 2, 0.0025 ** 2, 0.0025 ** 2],
    Q=0.2 ** 2,
    landmarks=[[3, -3, 0, 0, -3, 3, -3, 3], [0, 0, 2, -2, 2, 2, -2, -2]],
    cov0=[[0.001, 0, 0], [0, 0.001, 0], [0, 0, 0.001]],
)


class RefLoop:
    """oracle/_ref/libpocs_ref_loop.so: the reference's own estimator loop (MCSimulator.h's runSimulation() /
    runGMMEstimation(), compiled from the header by `make ref_loop`; ref_loop_harness.cpp) with the oracle's 2-D
    collision predicate in place of the one OpenRAVE call.  Test infrastructure (tests/test_oracle_vs_ref_loop.py)
    and bench.py's cpu_baseline leg."""
    LIB = HERE / "_ref" / "libpocs_ref_loop.so"

    def __init__(self, orc, pocs, plan, env):
        self.lib = C.CDLL(str(self.LIB))
        self.lib.refl2_run_mc.restype = C.c_double
        self.lib.refl2_run_gmm.restype = C.c_double
        self.lib.refl2_last_text.restype = C.c_longlong
        self.lib.refl2_last_error.restype = C.c_longlong
        self.lib.refl2_last_error.restype = C.c_longlong
        self.orc, self.pocs, self.plan, self.env = orc, pocs, plan, env

    def configure(self, particles, gaussians, samples, params=None):
        """params: dict(alphas, Q, landmarks 2 x L, cov0 3 x 3) in place of the reference's defaults."""
        d = params or (DEFAULTS if self.pocs is None else self.pocs.DEFAULTS)
        lm = np.asarray(d["landmarks"], np.float64)
        self.cfg = self.orc.config(self.plan, self.env, K=gaussians, alphas=d["alphas"], Q=d["Q"], landmarks=lm, cov0=d["cov0"])
        traj = np.ascontiguousarray(np.asarray(self.plan["traj"], np.float64).T)
        odom = np.ascontiguousarray(np.asarray(self.plan["odom"], np.float64).T)
        self.W, self.L, self.N, self.K = traj.shape[1], lm.shape[1], particles, gaussians
        self.samples = samples
        fn = C.cast(self.orc.lib.orc_collides_cfg, C.c_void_p)
        self.lib.refl2_configure(_p(np.asarray(d["alphas"], np.float64)), C.c_double(d["Q"]), _p(np.ascontiguousarray(lm[0])),
                                 _p(np.ascontiguousarray(lm[1])), C.c_int(self.L), _p(traj), _p(odom), C.c_int(self.W),
                                 _p(np.ascontiguousarray(np.asarray(d["cov0"], np.float64))), C.c_int(particles),
                                 C.c_int(gaussians), C.c_int(samples), fn, C.byref(self.cfg))
        return self.cfg

    def run_mc(self, seed):
        mu, cov, parts = np.zeros(3), np.zeros(9), np.zeros((self.N, 3))
        hits, checked = np.zeros(self.N, np.uint32), C.c_longlong(0)
        p = self.lib.refl2_run_mc(C.c_uint(seed), _p(mu), _p(cov), _p(parts), hits.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(checked))
        init, chain = np.zeros((self.N, 3)), np.zeros((self.W - 1, 3 + self.L))
        self.lib.refl2_record_mc(C.c_uint(seed), C.c_int(self.N), C.c_int(self.W), C.c_int(self.L), _p(init), _p(chain))
        return dict(p=p, mu=mu, cov=cov, particles=parts, hits=hits, checked=checked.value, init=init, chain=chain)

    def time_mc(self, seed):
        """runSimulation() alone (no tape): for timing."""
        mu, cov, parts = np.zeros(3), np.zeros(9), np.zeros((self.N, 3))
        hits, checked = np.zeros(self.N, np.uint32), C.c_longlong(0)
        return self.lib.refl2_run_mc(C.c_uint(seed), _p(mu), _p(cov), _p(parts), hits.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(checked))

    def run_gmm(self, seed, gen_seed=7, record=False):
        mu, cov = np.zeros(3), np.zeros(9)
        means, covs, checked = np.zeros((self.K, 3)), np.zeros((self.K, 9)), C.c_longlong(0)
        p = self.lib.refl2_run_gmm(C.c_uint(seed), C.c_uint(gen_seed), _p(mu), _p(cov), _p(means), _p(covs), C.byref(checked))
        if p != p:                                   # the reference's run ended in an exception (it has no handling of its own)
            return dict(p=p, error=self.last_error(), checked=checked.value)
        text = self.last_text()
        out = dict(p=p, mu=mu, cov=cov, means=means, covs=covs, checked=checked.value, probs=self.printed_probabilities(text),
                   counts=self.printed_counts(text))
        if record:
            gmm, chain = np.zeros((self.W, self.samples, 3)), np.zeros((self.W - 1, 3 + self.L))
            cnt = np.ascontiguousarray(out["counts"], np.int64)
            self.lib.refl2_record_gmm(C.c_uint(seed), C.c_int(self.samples), C.c_int(self.W), C.c_int(self.L), C.c_int(self.K),
                                      cnt.ctypes.data_as(C.POINTER(C.c_longlong)), _p(gmm), _p(chain))
            out.update(gmm=gmm, chain=chain)
        return out

    def last_error(self):
        """what() of the exception that ended the last run ("" if none): the reference has no handling of its own."""
        n = self.lib.refl2_last_error(None, C.c_longlong(0))
        buf = C.create_string_buffer(n + 1)
        self.lib.refl2_last_error(buf, C.c_longlong(n + 1))
        return buf.value.decode("ascii", "replace")

    def last_text(self):
        n = self.lib.refl2_last_text(None, C.c_longlong(0))
        buf = C.create_string_buffer(n + 1)
        self.lib.refl2_last_text(buf, C.c_longlong(n + 1))
        return buf.value.decode("ascii", "replace")

    def printed_counts(self, text):
        """The samples per component of every waypoint, as sampleNPoints prints them ("Counts Vector", GM_Model.h:95-96)."""
        rows = [[int(t) for t in m.split()] for m in re.findall(r"Counts Vector\n([0-9 ]+)\n", text)]
        assert len(rows) == self.W and all(len(r) == self.K and sum(r) == self.samples for r in rows), rows[:3]
        return np.array(rows, np.int64)

    def printed_probabilities(self, text):
        """The row the loop prints after "Collision Probabilities:" (MCSimulator.h:845-846; Armadillo's four decimals)."""
        m = re.search(r"Collision Probabilities:\n(.*?)\nCollision Free Probabilities:", text, re.S)
        assert m, text[-2000:]
        return np.array([float(t) for t in m.group(1).split()])
