"""ctypes binding of the C ABI declared in include/pocs.h (libpocs.so).

Nothing here computes: every call goes into the HIP library.  If libpocs.so is missing it is
built with hipcc (build.py); if that is impossible, or no GPU is usable, the error is raised --
there is no CPU fallback.
"""
import ctypes as C
from pathlib import Path

import numpy as np

from . import build as _build

POCS_OK = 0
E_ARG, E_ORDER, E_STATE, E_DEVICE, E_UNKNOWN_COMMAND, E_BUFFER = -1, -2, -3, -4, -5, -6
OPT_STORE_SAMPLES, OPT_MC_FUSED, OPT_USE_GRAPH, OPT_PROFILE, OPT_RUN_AHEAD, OPT_PERSISTENT, OPT_LONE_CALL = 1, 2, 3, 4, 5, 6, 7
OPT_SUB_BATCHES, OPT_MC_NONTEMPORAL = 8, 9
NMOM = 11

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); the authoritative list is include/pocs.h (tests check both agree)
SIGNATURES = {
    "pocs_create": (C.c_int, [C.POINTER(_vp), C.c_int]),
    "pocs_destroy": (None, [_vp]),
    "pocs_last_error": (C.c_char_p, [_vp]),
    "pocs_version": (C.c_char_p, []),
    "pocs_set_footprint": (C.c_int, [_vp, C.c_double, C.c_double, C.c_double, C.c_double]),
    "pocs_set_obstacles": (C.c_int, [_vp, _dp, C.c_int]),
    "pocs_set_alphas": (C.c_int, [_vp, _dp, C.c_int]),
    "pocs_set_q": (C.c_int, [_vp, C.c_double]),
    "pocs_set_num_landmarks": (C.c_int, [_vp, C.c_int]),
    "pocs_set_landmarks": (C.c_int, [_vp, _dp, C.c_int]),
    "pocs_set_num_particles": (C.c_int, [_vp, C.c_longlong]),
    "pocs_set_initial_covariance": (C.c_int, [_vp, _dp]),
    "pocs_set_path_length": (C.c_int, [_vp, C.c_int]),
    "pocs_set_trajectory": (C.c_int, [_vp, _dp, C.c_int]),
    "pocs_set_odometry": (C.c_int, [_vp, _dp, C.c_int]),
    "pocs_set_num_gaussians": (C.c_int, [_vp, C.c_int]),
    "pocs_set_num_gmm_samples": (C.c_int, [_vp, C.c_longlong]),
    "pocs_set_seed": (C.c_int, [_vp, C.c_uint64]),
    "pocs_run_simulation": (C.c_int, [_vp, _dp]),
    "pocs_run_gmm_estimation": (C.c_int, [_vp, _dp]),
    "pocs_send_command": (C.c_int, [_vp, C.c_char_p, C.c_char_p, C.c_size_t]),
    "pocs_set_option": (C.c_int, [_vp, C.c_int, C.c_longlong]),
    "pocs_set_batch": (C.c_int, [_vp, C.c_int]),
    "pocs_get_batch_probabilities": (C.c_int, [_vp, _dp, C.c_int]),
    "pocs_select_batch_run": (C.c_int, [_vp, C.c_int]),
    "pocs_set_shard": (C.c_int, [_vp, C.c_longlong, C.c_longlong]),
    "pocs_set_stream": (C.c_int, [_vp, _vp]),
    "pocs_gmm_begin": (C.c_int, [_vp]),
    "pocs_gmm_step_local": (C.c_int, [_vp, C.c_int]),
    "pocs_gmm_advance_local": (C.c_int, [_vp, C.c_int]),
    "pocs_gmm_sample_local": (C.c_int, [_vp, C.c_int]),
    "pocs_gmm_moments_ptr": (_vp, [_vp, C.c_int]),
    "pocs_gmm_moments_len": (C.c_int, [_vp]),
    "pocs_gmm_bind_moments": (C.c_int, [_vp, _vp, C.c_longlong]),
    "pocs_gmm_end": (C.c_int, [_vp, _dp]),
    "pocs_mc_run_local": (C.c_int, [_vp, C.POINTER(C.c_ulonglong)]),
    "pocs_mc_get_batch_counts": (C.c_int, [_vp, C.POINTER(C.c_ulonglong), C.c_int]),
    "pocs_xchg_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_void_p]),
    "pocs_xchg_connect": (C.c_int, [_vp, C.c_void_p, C.c_int]),
    "pocs_gmm_exchange_local": (C.c_int, [_vp, C.c_int]),
    "pocs_gmm_sample_exchange_local": (C.c_int, [_vp, C.c_int]),
    "pocs_get_path_length": (C.c_int, [_vp]),
    "pocs_get_waypoint_probabilities": (C.c_int, [_vp, _dp, C.c_int]),
    "pocs_get_moments": (C.c_int, [_vp, C.c_int, _dp, C.c_int]),
    "pocs_get_gmm_state": (C.c_int, [_vp, C.c_int, _dp, _dp, _dp, _dp]),
    "pocs_get_host_chain": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp]),
    "pocs_copy_gmm_samples": (C.c_longlong, [_vp, _dp, C.POINTER(C.c_int16), C.c_longlong]),
    "pocs_copy_particles": (C.c_longlong, [_vp, _dp, C.POINTER(C.c_uint32), C.c_longlong]),
    "pocs_measure_copy_bandwidth": (C.c_int, [_vp, C.c_longlong, _dp]),
    "pocs_measure_fill_bandwidth": (C.c_int, [_vp, C.c_longlong, _dp]),
    "pocs_get_kernel_time": (C.c_int, [_vp, _dp, C.POINTER(C.c_longlong)]),
    "pocs_get_sequence_time": (C.c_int, [_vp, _dp, C.POINTER(C.c_int)]),
    "pocs_get_exchange_wait": (C.c_int, [_vp, _dp]),
    "pocs_probe_device_math": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), _dp, _dp, _dp, _dp, _dp, _dp]),
}

_lib = None


def library_path():
    return _build.LIB


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  A PyTorch-ROCm wheel carries its own libamdhip64.so (soname
    libamdhip64.so.7, the same as /opt/rocm's) and asks for it by FILE name: imported after libpocs.so
    has pulled in /opt/rocm/lib/libamdhip64.so.7 it loads a second runtime, whose hipInit then finds
    "no ROCm-capable device".  The other order is fine (libpocs.so asks by soname and binds to the copy
    already loaded), so where such a wheel is installed its runtime is mapped first -- without importing
    torch.  No torch: nothing happens, the system runtime serves libpocs.so."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for loc in (spec.submodule_search_locations or []) if spec else []:
        rt = Path(loc) / "lib" / "libamdhip64.so"
        if rt.exists():
            try:
                C.CDLL(str(rt), mode=C.RTLD_GLOBAL)
            except OSError:
                pass
            return


def load_library(build=True):
    """dlopen libpocs.so (building it first if needed) and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    import os
    path = os.environ.get("POCS_LIB")          # tuning only: an alternative build of the same library
    if not path:
        if build:
            _build.build_library()
        path = str(_build.LIB)
    if not Path(path).exists():
        raise RuntimeError("libpocs.so is missing and was not built; the HIP path is the only path")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        if os.environ.get("POCS_LIB") and not hasattr(lib, name):
            continue                 # tuning only: an A/B build from an older commit may predate an entry point
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class PocsError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("pocs error %d: %s" % (code, text))
        self.code = code


def _arr(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Context:
    """One estimator context on one GPU (the counterpart of one MCModule + its MCSimulator)."""

    def __init__(self, device=0):
        self.lib = load_library()
        h = _vp()
        rc = self.lib.pocs_create(C.byref(h), device)
        self.h = h
        if rc != POCS_OK:
            msg = self.lib.pocs_last_error(h).decode() if h else "allocation failed"
            if h:
                self.lib.pocs_destroy(h)
            self.h = None
            raise PocsError(rc, msg)

    def close(self):
        if getattr(self, "h", None):
            self.lib.pocs_destroy(self.h)
            self.h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise PocsError(rc, self.lib.pocs_last_error(self.h).decode())
        return rc

    # ---- text channel (MCModule.SendCommand) ------------------------------------------
    def send_command(self, line, cap=1 << 16):
        buf = C.create_string_buffer(cap)
        self._chk(self.lib.pocs_send_command(self.h, line.encode(), buf, cap))
        return buf.value.decode()

    SendCommand = send_command

    # ---- typed setters ----------------------------------------------------------------
    def set_env(self, env):
        fp = env["footprint"]
        self._chk(self.lib.pocs_set_footprint(self.h, fp[0], fp[1], fp[2], fp[3]))
        if env.get("boxes") is None:        # footprint only: the world itself is left as it is
            return
        b = _arr(env["boxes"]).reshape(-1, 5)
        self._chk(self.lib.pocs_set_obstacles(self.h, b.ctypes.data_as(_dp), b.shape[0]))

    def set_alphas(self, a):
        a = _arr(a)
        self._chk(self.lib.pocs_set_alphas(self.h, a.ctypes.data_as(_dp), len(a)))

    def set_q(self, q):
        self._chk(self.lib.pocs_set_q(self.h, q))

    def set_landmarks(self, lm):
        lm = _arr(lm)                       # 2 x L: xs then ys
        self._chk(self.lib.pocs_set_num_landmarks(self.h, lm.shape[1]))
        self._chk(self.lib.pocs_set_landmarks(self.h, lm.ctypes.data_as(_dp), lm.shape[1]))

    def set_num_particles(self, n):
        self._chk(self.lib.pocs_set_num_particles(self.h, n))

    def set_initial_covariance(self, c):
        c = _arr(c).ravel()
        self._chk(self.lib.pocs_set_initial_covariance(self.h, c.ctypes.data_as(_dp)))

    def set_plan(self, plan):
        traj = _arr(np.asarray(plan["traj"]).T)     # by component, as setTrajectory expects
        odom = _arr(np.asarray(plan["odom"]).T)
        W = traj.shape[1]
        self._chk(self.lib.pocs_set_path_length(self.h, W))
        self._chk(self.lib.pocs_set_trajectory(self.h, traj.ctypes.data_as(_dp), W))
        self._chk(self.lib.pocs_set_odometry(self.h, odom.ctypes.data_as(_dp), W - 1))

    def set_num_gaussians(self, k):
        self._chk(self.lib.pocs_set_num_gaussians(self.h, k))

    def set_num_gmm_samples(self, n):
        self._chk(self.lib.pocs_set_num_gmm_samples(self.h, n))

    def set_seed(self, s):
        self._chk(self.lib.pocs_set_seed(self.h, s))

    def set_option(self, opt, val):
        self._chk(self.lib.pocs_set_option(self.h, opt, val))

    def set_batch(self, runs):
        """Independent GMM estimations advanced in lockstep per run_gmm_estimation / begin..end."""
        self._chk(self.lib.pocs_set_batch(self.h, runs))
        self._batch = runs

    def batch_probabilities(self):
        n = getattr(self, "_batch", 1)
        out = np.zeros(n)
        got = self._chk(self.lib.pocs_get_batch_probabilities(self.h, out.ctypes.data_as(_dp), n))
        return out[:got]

    def select_batch_run(self, run):
        """The getters (waypoint probabilities, moments, mixture state, samples ...) expose run `run` of the last batch."""
        self._chk(self.lib.pocs_select_batch_run(self.h, int(run)))

    def set_shard(self, first=-1, count=-1):
        """Evaluate global indices [first, first+count); no arguments = the whole range."""
        self._chk(self.lib.pocs_set_shard(self.h, first, count))

    def set_stream(self, stream_ptr):
        self._chk(self.lib.pocs_set_stream(self.h, stream_ptr))

    def configure(self, plan, env, params=None, K=None, N=None, seed=None):
        """Everything MCSimulation.py:154-207 pushes, from Python values."""
        from .planio import DEFAULTS
        p = dict(DEFAULTS)
        p.update(params or {})
        self.set_env(env)
        self.set_alphas(p["alphas"])
        self.set_q(p["Q"])
        self.set_landmarks(p["landmarks"])
        n = p["num_particles"] if N is None else N
        self.set_num_particles(n)
        self.set_initial_covariance(p["cov0"])
        self.set_plan(plan)
        self.set_num_gaussians(p["num_gaussians"] if K is None else K)
        self.set_num_gmm_samples(n)
        if seed is not None:
            self.set_seed(seed)

    # ---- runs -------------------------------------------------------------------------
    def run_simulation(self):
        p = C.c_double()
        self._chk(self.lib.pocs_run_simulation(self.h, C.byref(p)))
        return p.value

    def run_gmm_estimation(self):
        p = C.c_double()
        self._chk(self.lib.pocs_run_gmm_estimation(self.h, C.byref(p)))
        return p.value

    def mc_run_local(self):
        n = C.c_ulonglong()
        self._chk(self.lib.pocs_mc_run_local(self.h, C.byref(n)))
        return n.value

    def mc_batch_counts(self):
        n = getattr(self, "_batch", 1)
        out = (C.c_ulonglong * n)()
        got = self._chk(self.lib.pocs_mc_get_batch_counts(self.h, out, n))
        return [int(v) for v in out[:got]]

    def gmm_begin(self):
        self._chk(self.lib.pocs_gmm_begin(self.h))

    def gmm_step_local(self, w):
        self._chk(self.lib.pocs_gmm_step_local(self.h, w))

    def gmm_advance_local(self, w):
        self._chk(self.lib.pocs_gmm_advance_local(self.h, w))

    def gmm_sample_local(self, w):
        self._chk(self.lib.pocs_gmm_sample_local(self.h, w))

    def xchg_create(self, world, rank):
        """This rank's exchange buffer; returns its 64-byte IPC handle (bytes) for the peers."""
        h = C.create_string_buffer(64)
        self._chk(self.lib.pocs_xchg_create(self.h, world, rank, C.cast(h, C.c_void_p)))
        return h.raw

    def xchg_connect(self, handles):
        """handles: the 64-byte handles of rank 0 .. world-1, in rank order."""
        blob = b"".join(handles)
        buf = C.create_string_buffer(blob, len(blob))
        self._chk(self.lib.pocs_xchg_connect(self.h, C.cast(buf, C.c_void_p), len(handles)))

    def gmm_exchange_local(self, w):
        self._chk(self.lib.pocs_gmm_exchange_local(self.h, w))

    def gmm_sample_exchange_local(self, w):
        """sample(w) and exchange(w) in one launch (include/pocs.h)."""
        self._chk(self.lib.pocs_gmm_sample_exchange_local(self.h, w))

    def gmm_moments_ptr(self, w):
        return self.lib.pocs_gmm_moments_ptr(self.h, w)

    def gmm_moments_len(self):
        return self.lib.pocs_gmm_moments_len(self.h)

    def gmm_bind_moments(self, ptr, n):
        self._chk(self.lib.pocs_gmm_bind_moments(self.h, ptr, n))

    def gmm_end(self):
        p = C.c_double()
        self._chk(self.lib.pocs_gmm_end(self.h, C.byref(p)))
        return p.value

    # ---- results ----------------------------------------------------------------------
    def path_length(self):
        return self.lib.pocs_get_path_length(self.h)

    def waypoint_probabilities(self):
        W = self.path_length()
        out = np.zeros(W)
        n = self._chk(self.lib.pocs_get_waypoint_probabilities(self.h, out.ctypes.data_as(_dp), W))
        return out[:n]

    def moments(self, w, K):
        out = np.zeros(K * NMOM)
        self._chk(self.lib.pocs_get_moments(self.h, w, out.ctypes.data_as(_dp), out.size))
        return out.reshape(K, NMOM)

    def gmm_state(self, w, K):
        m, c, wt, al = np.zeros((K, 3)), np.zeros((K, 9)), np.zeros(K), np.zeros(K)
        self._chk(self.lib.pocs_get_gmm_state(self.h, w, m.ctypes.data_as(_dp), c.ctypes.data_as(_dp),
                                              wt.ctypes.data_as(_dp), al.ctypes.data_as(_dp)))
        return m, c.reshape(K, 3, 3), wt, al

    def gmm_state_raw(self, w, K):
        """K x 16 rows [mean(3) cov(9) weight alive 0 0] -- the layout the oracle uses too."""
        m, c, wt, al = self.gmm_state(w, K)
        s = np.zeros((K, 16))
        s[:, 0:3], s[:, 3:12], s[:, 12], s[:, 13] = m, c.reshape(K, 9), wt, al
        return s

    def host_chain(self, L):
        n = max(self.path_length() - 1, 0)
        a, no, z = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, L))
        mu, cov = np.zeros((n, 3)), np.zeros((n, 9))
        self._chk(self.lib.pocs_get_host_chain(self.h, *(x.ctypes.data_as(_dp) for x in (a, no, z, mu, cov))))
        return dict(applied=a, noisy=no, z=z, mu=mu, cov=cov)

    def gmm_samples(self, n):
        xyz = np.zeros((n, 3))
        flags = np.zeros(n, np.int16)
        got = self._chk(self.lib.pocs_copy_gmm_samples(self.h, xyz.ctypes.data_as(_dp),
                                                       flags.ctypes.data_as(C.POINTER(C.c_int16)), n))
        return xyz[:got], flags[:got]

    def particles(self, n):
        xyz = np.zeros((n, 3))
        hits = np.zeros(n, np.uint32)
        got = self._chk(self.lib.pocs_copy_particles(self.h, xyz.ctypes.data_as(_dp),
                                                     hits.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return xyz[:got], hits[:got]

    def copy_bandwidth(self, nbytes=1 << 30):
        """GB/s (read + written) of a plain streaming copy of nbytes on this GPU."""
        g = C.c_double()
        self._chk(self.lib.pocs_measure_copy_bandwidth(self.h, nbytes, C.byref(g)))
        return g.value

    def fill_bandwidth(self, nbytes=1 << 30):
        """GB/s written by a plain streaming fill of nbytes on this GPU."""
        g = C.c_double()
        self._chk(self.lib.pocs_measure_fill_bandwidth(self.h, nbytes, C.byref(g)))
        return g.value

    def sequence_time(self):
        """(ms from the first sampling launch to the end of the last, sub-batches in flight side by side) of the last
        whole-run GMM call under OPT_PROFILE."""
        ms, g = C.c_double(0.0), C.c_int(1)
        self._chk(self.lib.pocs_get_sequence_time(self.h, C.byref(ms), C.byref(g)))
        return ms.value, g.value

    def exchange_wait_us(self):
        """(min, median, max) over the (run, waypoint) pairs of the last sharded begin..end sequence of how long the
        closers waited for the other ranks' moments, in microseconds (the library's own exchange only)."""
        v = (C.c_double * 3)()
        self._chk(self.lib.pocs_get_exchange_wait(self.h, v))
        return v[0], v[1], v[2]

    def probe_device_math(self, radius_words, angle_words, headings):
        """Test hook: the device's table-driven sampler functions on chosen inputs -> (z0, z1, sin, cos, radius^2) arrays."""
        wr = np.ascontiguousarray(radius_words, dtype=np.uint32)
        wa = np.ascontiguousarray(angle_words, dtype=np.uint32)
        x = np.ascontiguousarray(headings, dtype=np.float64)
        n = len(wr)
        assert len(wa) == n and len(x) == n
        out = [np.empty(n) for _ in range(5)]
        u32 = C.POINTER(C.c_uint32)
        self._chk(self.lib.pocs_probe_device_math(self.h, n, wr.ctypes.data_as(u32), wa.ctypes.data_as(u32), x.ctypes.data_as(_dp),
                                                  *[o.ctypes.data_as(_dp) for o in out]))
        return tuple(out)

    def kernel_time(self):
        ms, n = C.c_double(), C.c_longlong()
        self._chk(self.lib.pocs_get_kernel_time(self.h, C.byref(ms), C.byref(n)))
        return ms.value, n.value
