#!/usr/bin/env python3
"""bench.py -- particle-waypoint evaluations/s of the GMM + collision hot path on MI355X.

One "step" = one runGMMEstimation over the whole plan (W waypoints x N samples) on synthetic
(seeded) draws -- one iteration of the reference driver's 200-run loop (MCSimulation.py:238-256).
Steps are independent, so the GPU advances `--batch` of them in lockstep per call (one launch per
waypoint for the whole batch, pocs_set_batch); K steps = K/batch calls (+ one call with the
remainder).  Default workload at --gpus 1 is BASELINE.json configs[1]: the bundled
trajectory.dat / odometry.dat plan (56 waypoints), 10^6 samples, 3-component mixture.

With --gpus G (launched by torch.distributed.run, one rank per GPU) every rank evaluates its own
10^6 samples of a G x 10^6-sample mixture (weak scaling, the default) or its 1/G of the workload's
samples (--scaling strong: BASELINE.json configs[3] = `--workload cfg3 --scaling strong --gpus 8`);
per waypoint the moments of the whole batch (batch x 11 K doubles) are summed over ranks with ONE
RCCL all-reduce.

Timing: after the warm-up the K-step pass is repeated -- at least 10 times and for about a second
of GPU time, each repeat bracketed by barrier + synchronize on both sides and taken as the MAX over ranks --
and `ms_per_step` / `value` are the MEDIAN repeat's; min / max / repeats travel in `timing`.

Prints ONE JSON line on rank 0 (fields: DESIGN.md section 7).
"""
import argparse
import json
import math
import os
import statistics
import sys
import time
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
NUMERICS = "v9"                 # counter records of another numerics version describe another kernel
BYTES_PER_EVAL_GMM = 26         # 3 x f64 sample + i16 flag streamed out once
BYTES_PER_EVAL_MC = 56          # 24 B in + 24 B out + u32 hit counter read + write
CPU_BUDGET_FACTOR = 3           # the CPU baseline may take this many times the wall time of the run's GPU part
TIMED_TARGET_S = float(os.environ.get("POCS_BENCH_TARGET_S", "1.5"))   # GPU time the repeats of the timed pass add up to
MC_CACHE_BYTES = 232.0e6        # particle state of a batch up to this size stays in the 256 MB Infinity Cache (pocs_host.hip)

WORKLOADS = {
    # name: (waypoints, samples per GPU, components, path)
    "cfg2": (56, 1_000_000, 3, "gmm"),      # bundled plan, 1M samples, K=3   <- the metric's config
    "cfg3": (500, 10_000_000, 8, "gmm"),    # 500-waypoint resampled plan, 10M samples, K=8
    "cfg5": (500, 100_000, 1, "mc"),        # MC roll-outs, 500 waypoints
    "mc": (56, 1_000_000, 1, "mc"),
}


def cpu_baseline(plan, env, K, W, path, budget_evals, budget_s=None):
    """The reference's own loop (compiled from MCSimulator.h into oracle/_ref where that travelled with the tree) and
    the oracle (single-thread C restatement) timed on this host on a bounded sample: 1 thread (the reference has no
    threading), then the restatement on one GPU's share of the host (16 threads) and on EVERY core the process may use,
    one independent run per thread (the reference's 200 runs are independent).  budget_s: wall seconds the whole
    baseline may take -- the caller passes a stated multiple of what the GPU part took, so that the GPU part is not a
    blip in the run; the sample is sized from a short calibration run."""
    import threading
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle
    orc = oracle.Oracle()
    cfg = orc.config(plan, env, K=K)

    def one(seed, n):
        if path == "gmm":
            orc.run_gmm(cfg, seed, n)
        else:
            orc.run_mc(cfg, seed, n)

    n = max(1000, int(budget_evals // W))
    legs = 4 if oracle.RefLoop.LIB.exists() else 3
    if budget_s is not None:
        t0 = time.perf_counter()
        one(99, 2000)
        rate = 2000 * W / (time.perf_counter() - t0)           # evaluations per second of one thread, roughly
        n = max(1000, min(n, int(0.8 * rate * budget_s / legs / W)))
    t0 = time.perf_counter()
    one(1234, n)
    dt = time.perf_counter() - t0
    out = {"value": n * W / dt, "unit": "particle-waypoint evals/s", "cores": 1, "kind": "port",
           "sample": "%d samples x %d waypoints (%s path, K=%d), oracle/pocs_oracle.c, 1 thread, %.1f s"
                     % (n, W, path, K, dt)}
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out["nproc"], out["cpu_model"] = ncpu, model
    if budget_s is not None:
        out["budget"] = "the whole baseline sized to <= %.1f s = %s x the %.1f s the GPU part of this run took" % (
            budget_s, CPU_BUDGET_FACTOR, budget_s / CPU_BUDGET_FACTOR)
    # the reference's OWN loop where its compiled pieces travelled with the tree (oracle/_ref, built by `make -C oracle
    # ref_loop` from MCSimulator.h where it lies; nothing of /root/reference is read here): runGMMEstimation() /
    # runSimulation() on the same plan, world and sample.  It then IS the baseline ("kind": "reference") and the
    # restatement's figures move under "port".  Its collision check is this build's cheap 2-D predicate, not
    # OpenRAVE's mesh query, which dominates the reference's real cost: an upper bound of the reference's speed.
    if oracle.RefLoop.LIB.exists():
        try:
            ref = oracle.RefLoop(orc, None, plan, env)
            ref.configure(particles=n if path == "mc" else 10, gaussians=K, samples=n if path == "gmm" else 10)
            t0 = time.perf_counter()
            if path == "gmm":
                res = ref.run_gmm(4321, gen_seed=8765)
                p_ref, err = res["p"], res.get("error")
            else:
                p_ref, err = ref.time_mc(4321), None
            dtr = time.perf_counter() - t0
            # the reference has no error handling of its own: a run that threw (a Gaussian that lost all its samples)
            # ended EARLY and returns NaN -- its time is not the time of a run, and the restatement's figure stands
            if err or p_ref != p_ref:
                out["reference_error"] = ("the reference's loop threw: %s" % (err or ref.last_error()))[:200]
            else:
                port = {k: out[k] for k in ("value", "cores", "kind", "sample")}
                out.update({"value": n * W / dtr, "cores": 1, "kind": "reference", "port": port, "reference_probability": p_ref,
                            "sample": "%d samples x %d waypoints (%s path, K=%d): the reference's own %s compiled from "
                                      "MCSimulator.h (oracle/_ref/libpocs_ref_loop.so), its one OpenRAVE collision call "
                                      "replaced by this build's 2-D predicate, 1 thread, %.1f s"
                                      % (n, W, path, K, "runGMMEstimation()" if path == "gmm" else "runSimulation()", dtr)})
        except Exception as e:                                                    # noqa: BLE001 -- the port stands
            out["reference_error"] = repr(e)[:200]

    def threaded(nthreads, n_each):
        th = [threading.Thread(target=one, args=(2000 + i, n_each)) for i in range(nthreads)]      # ctypes releases the GIL
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dta = time.perf_counter() - t0
        return {"value": nthreads * n_each * W / dta, "cores": nthreads, "kind": "port",
                "sample": "%d independent runs of %d samples of the restatement, one per thread, %.1f s" % (nthreads, n_each, dta)}

    share = min(ncpu, int(os.environ.get("POCS_CPU_THREADS", "16")))      # one GPU's share of the host on the pool
    if share > 1:
        out["all_cores"] = threaded(share, n)
        out["all_cores"]["note"] = "one GPU's share of this host (POCS_CPU_THREADS, default 16)"
    if ncpu > share:                                                      # ... and the whole host: every core the process may use
        # (the same total work as the leg above, spread over all of them: on the pool's boxes the affinity mask shows
        # the whole host while a cgroup quota holds the process to one GPU's share of it -- `cgroup_cpu_limit` says so,
        # and the figure then reads BELOW the 16-thread one: oversubscription, not a slower host)
        out["host_cores"] = threaded(ncpu, max(1000, n * share // ncpu))
        out["host_cores"]["note"] = "every core of the host this process may run on (sched_getaffinity)"
        try:
            quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            out["host_cores"]["cgroup_cpu_limit"] = None if quota == "max" else float(quota) / float(period)
        except (OSError, ValueError):
            out["host_cores"]["cgroup_cpu_limit"] = None
    return out


class BoardProbe:
    """Shader clock and socket power of one GPU while the kernels run, read from the amdgpu hwmon files
    (no subprocess): the GMM kernel runs at the board's power cap, and the clock the cap leaves it is
    part of the roofline's story.  Everything here is best effort: a missing file gives None."""

    def __init__(self, pci_bus_id=None):
        import glob
        self.dirs = []
        for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
            if os.path.exists(h + "/power1_input") and os.path.exists(h + "/freq1_input"):
                dev = os.path.realpath(h + "/../..")
                self.dirs.append((h, dev))
        if pci_bus_id:
            hit = [d for d in self.dirs if os.path.basename(d[1]).lower().endswith(pci_bus_id.lower())]
            self.dirs = hit or self.dirs
        self.samples = {h: [] for h, _ in self.dirs}
        self.thread, self.stop_flag = None, False

    @staticmethod
    def _num(path):
        try:
            with open(path) as f:
                return float(f.read().split()[0])
        except (OSError, ValueError, IndexError):
            return None

    def _loop(self):
        while not self.stop_flag:
            for h, _ in self.dirs:
                pw, fq = self._num(h + "/power1_input"), self._num(h + "/freq1_input")
                if pw is not None and fq is not None:
                    self.samples[h].append((pw * 1e-6, fq * 1e-6))
            time.sleep(0.02)

    def start(self):
        if self.dirs:
            import threading
            self.thread = threading.Thread(target=self._loop, daemon=True)
            self.thread.start()

    def stop(self):
        if self.thread is None:
            return None
        self.stop_flag = True
        self.thread.join()
        # several cards visible and no bus id to tell them apart: the busy one is the one drawing power
        best = max(self.samples.items(), key=lambda kv: sum(p for p, _ in kv[1]) / max(len(kv[1]), 1), default=None)
        if not best or not best[1]:
            return None
        h, smp = best
        dev = dict(self.dirs)[h]
        peak = None
        try:
            with open(dev + "/pp_dpm_sclk") as f:
                peak = max(float(t[:-3]) for t in f.read().replace("*", " ").split() if t.lower().endswith("mhz"))
        except (OSError, ValueError):
            pass
        cap = self._num(h + "/power1_cap")
        return {"sclk_MHz": sum(f for _, f in smp) / len(smp), "sclk_peak_MHz": peak,
                "power_W": max(p for p, _ in smp), "power_cap_W": cap * 1e-6 if cap else None, "samples": len(smp),
                "source": "amdgpu hwmon freq1_input (mean) / power1_input (max) polled for 0.4 s of further calls after all "
                          "timing; held for 10 s: tools/clock_probe.sh"}


def probe_onehop(par, pocs_amd, torch, dist, rank, world, local):
    """One small sharded GMM call through the library's IPC exchange -- the whole call in one library call, as the
    timed runs make it (a connected context's pocs_run_gmm_estimation: graph replay, exchange in the launches' tails) -- on
    this node: True if it ran on every rank and every rank got the same probability.  A node where peer
    mapping or in-kernel peer traffic does not work shows here (an exception, or the kernel's bounded wait
    giving up after 30 s), before anything is timed.  Every rank issues the SAME sequence of collectives
    whatever fails where: the ranks agree on ok / failed (an all_reduce of a status word) after the buffers
    exist, after the handles have gone round, and after the call."""
    dev = "cuda" if os.environ.get("POCS_DIST_BACKEND", "nccl") == "nccl" else "cpu"

    def agree(ok, extra=()):
        if dist is None:
            return bool(ok), list(extra)
        t = torch.tensor([1.0 if ok else 0.0] + [float(v) for v in extra], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t[0].item()), t[1:].tolist()

    ctx, mine, ok = None, None, True
    try:
        plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
        ctx = pocs_amd.Context(local)
        ctx.configure(plan, env, K=3, N=8192 * world, seed=0x5EED00AA)
        ctx.set_batch(2)
        ctx.set_shard(rank * 8192, 8192)
        mine = ctx.xchg_create(world, rank)
    except Exception as exc:                               # noqa: BLE001
        print("one-hop probe (setup): %s" % exc, file=sys.stderr)
        ok = False
    ok, _ = agree(ok)
    handles = [mine]
    if dist is not None and world > 1:                     # every rank takes part, with a placeholder if it has no buffer
        handles = [None] * world
        dist.all_gather_object(handles, mine if mine is not None else b"")
    if ok:
        try:
            ctx.xchg_connect(handles)
        except Exception as exc:                           # noqa: BLE001
            print("one-hop probe (connect): %s" % exc, file=sys.stderr)
            ok = False
    ok, _ = agree(ok)
    p = -1.0
    if ok:
        try:
            p = ctx.run_gmm_estimation()
            torch.cuda.synchronize()
        except Exception as exc:                           # noqa: BLE001 -- whatever it is, the other path is taken
            print("one-hop probe (call): %s" % exc, file=sys.stderr)
            ok = False
    ok, mm = agree(ok, (p, -p))                            # min ok, min p, -max p; also: nobody unmaps a buffer still written to
    if ctx is not None:
        try:
            ctx.close()
        except Exception:                                  # noqa: BLE001
            pass
    return ok and mm[0] == -mm[1]


def spawn_ranks(n):
    """`bench.py --gpus N` started by hand (no RANK / WORLD_SIZE in the environment): bring up the N ranks
    ourselves -- N fresh child processes of this script, one per GPU, before this process has made any GPU
    call -- relay rank 0's one JSON line, and fail if any rank fails.  (A launcher such as
    `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` sets RANK / WORLD_SIZE itself
    and never gets here.)"""
    import socket
    import subprocess
    forced = os.environ.get("POCS_FORCE_DEVICE")          # rehearsal: every rank on one card (gloo)
    if forced is None:
        try:
            import torch                                   # device_count() does not initialise the GPU on this image
            have = torch.cuda.device_count()
        except Exception:                                  # noqa: BLE001
            have = 0
        if have < n:
            print("bench.py: --gpus %d but this node shows %d GPU(s): refusing to print a %d-GPU line from fewer "
                  "(rehearse the multi-rank path on one card with POCS_FORCE_DEVICE=0 POCS_DIST_BACKEND=gloo)" % (n, have, n),
                  file=sys.stderr)
            return 2
    for attempt in range(3):                              # the port is free when probed; another process may take it before
        with socket.socket() as sk:                       # rank 0 listens: a rendezvous that fails that way is tried again
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for rank in range(n):
            env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            # rank 0's stdout (the one JSON line) is held back until every rank has succeeded; the others print nothing there
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL,
                                          stderr=subprocess.PIPE if rank == 0 else None, text=True))
        # poll: the first rank to fail ends the job at once -- the others would sit in their collectives until the device
        # wait (30 s) or the process group's timeout (minutes) otherwise
        import threading
        grabbed = {}
        t_out = threading.Thread(target=lambda: grabbed.update(out=procs[0].stdout.read()), daemon=True)
        t_err = threading.Thread(target=lambda: grabbed.update(err=procs[0].stderr.read()), daemon=True)
        t_out.start(); t_err.start()
        failed = None
        while failed is None and any(p.poll() is None for p in procs):
            for i, p in enumerate(procs):
                if p.poll() not in (None, 0):
                    failed = i
                    break
            time.sleep(0.05)
        if failed is None:
            failed = next((i for i, p in enumerate(procs) if p.returncode != 0), None)
        if failed is not None:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
        t_out.join(timeout=10); t_err.join(timeout=10)
        err = grabbed.get("err", "") or ""
        if failed is not None and attempt < 2 and ("EADDRINUSE" in err or "Address already in use" in err or "address already in use" in err):
            print("bench.py: rendezvous port %d was taken before rank 0 listened; trying another" % port, file=sys.stderr)
            continue
        sys.stderr.write(err)
        if failed is not None:
            print("bench.py: rank %d exited with %s; the other ranks were stopped (%s)" % (failed, procs[failed].returncode, [p.returncode for p in procs]), file=sys.stderr)
            return 1
        sys.stdout.write(grabbed.get("out", "") or "")
        sys.stdout.flush()
        return 0
    return 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0,
                    help="independent runs (steps) advanced in lockstep per call (pocs_set_batch); "
                         "default 64 for the GMM path (the per-launch tail -- reduce, mixture advance, launch gap, "
                         "~12 us -- is paid once per waypoint for the whole batch), 8 x 10^6 / particles for MC (8 x 28 MB of particle "
                         "state stay in the 256 MB Infinity Cache between waypoint launches, 16 x do not)")
    ap.add_argument("--samples", type=int, default=0, help="override samples per GPU (experiments only)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's sample count PER GPU; strong: the workload's sample count in total, "
                         "split evenly over the GPUs (cfg4 = cfg3 over 8 GPUs)")
    ap.add_argument("--mc-fused", action="store_true",
                    help="MC workloads: whole roll-out in registers (k_mc_fused, ~0 B/eval) instead of streaming")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strong-record", action="store_true",
                    help="N > 1: skip the short strong-scaling pass (cfg3 split over the ranks = BASELINE configs[3] at N = 8) "
                         "that the default weak line carries as `strong`")
    ap.add_argument("--cpu-evals", type=float, default=6.0e7, help="upper bound of the CPU baseline's sample (evaluations)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))          # this process never touches the GPU: it starts the ranks and relays

    # stdout carries the ONE JSON line and nothing else: libraries that write to fd 1 (RCCL prints
    # a version banner at communicator creation) go to stderr until the line is printed
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    t_process = time.perf_counter()

    import torch
    import pocs_amd
    par = import_module("probability-of-collision-for-safe-planning_amd.parallel")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if "POCS_FORCE_DEVICE" in os.environ:            # rehearsal: several ranks on one card (gloo)
        local = int(os.environ["POCS_FORCE_DEVICE"])
    backend = os.environ.get("POCS_DIST_BACKEND", "nccl")
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dist = None
    sharded = world > 1 or os.environ.get("POCS_FORCE_SHARDED") == "1"   # rehearse the N>1 path on one GPU
    if sharded:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))     # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    coll_dev = "cuda" if backend == "nccl" else "cpu"
    env = pocs_amd.load_env()
    t_process = time.perf_counter()                    # (from here on the GPU works: what the CPU baseline's budget is a multiple of)

    def over_ranks(values, op):
        """all_reduce of a few doubles (MAX / MIN / SUM); the identity on one rank."""
        if dist is None:
            return list(values)
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=op)
        return t.tolist()

    def gather_ranks(value):
        if dist is None:
            return [float(value)]
        t = torch.zeros(world, dtype=torch.float64, device=coll_dev)
        t[rank] = float(value)
        dist.all_reduce(t)
        return t.tolist()

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # How the shards of the GMM path exchange their moments (POCS_ONEHOP): "2" = the WHOLE CALL in one library call of a
    # context connected to its peers -- the same replayed graph and the same two sub-batches as on one GPU, every run's
    # moments exchanged over one hop in the closing block of its launch, no host in the loop (the default for N > 1,
    # after a small end-to-end probe of it on this node has succeeded on every rank); "3" = the same exchange with the
    # launches issued one waypoint at a time from here (pocs_gmm_sample_exchange_local: round 3's default), "1" = the
    # one-hop exchange as its own launch, "0" = one RCCL all-reduce per waypoint (two engines).
    xstate = {"mode": os.environ.get("POCS_ONEHOP", "2" if (sharded and WORKLOADS[args.workload][3] == "gmm") else "0"),
              "note": None, "probe": None}
    if sharded and xstate["mode"] == "2" and "POCS_ONEHOP" not in os.environ:
        t0 = time.perf_counter()
        ok = probe_onehop(par, pocs_amd, torch, dist if world > 1 else None, rank, world, local)
        xstate["probe"] = {"ok": bool(ok), "seconds": time.perf_counter() - t0,
                           "what": "one small sharded GMM call through the library's IPC exchange on this node, every rank, before anything is timed"}
        if not ok:
            xstate["mode"], xstate["note"] = "0", "one-hop probe failed on this node: fell back to RCCL"
    os.environ["POCS_ONEHOP"] = xstate["mode"]

    def measure(workload, scaling, steps, warmup, batch_arg, samples, full):
        """One workload through the timing protocol; full = the main record (roofline extras, single call, board)."""
        W, n_local, K, path = WORKLOADS[workload]
        if samples:
            n_local = samples
        if scaling == "strong":           # the same total workload over more GPUs (even shards: pairs of samples share draws)
            n_local = max(2, (n_local // world) & ~1)
        plan = pocs_amd.load_plan()
        if W != 56:
            plan = pocs_amd.resample_plan(plan, W)
        N = n_local * world
        xmode = xstate["mode"] if path == "gmm" else "0"

        # K steps are issued as `ncalls` calls of nearly equal batch: `n_hi` calls of b_hi = b_lo + 1
        # runs and the rest of b_lo runs.
        # MC: as many roll-outs per launch as keep 8 x 10^6 particles' state (28 B each) in flight -- what stays in
        # the 256 MB Infinity Cache between two waypoint launches: 8 at 10^6 particles, 64 at 10^5 (cfg5: 0.41 -> 0.75)
        maxb = batch_arg if batch_arg > 0 else (64 if path == "gmm" else max(1, min(64, int(8_000_000 // max(n_local, 1)))))
        ncalls = (steps + maxb - 1) // maxb
        fused_exchange = xmode in ("2", "3")
        whole = sharded and path == "gmm" and xmode == "2"       # a connected context's whole-run call: the unsharded code path below
        if sharded and path == "gmm" and steps >= 2 and not fused_exchange:
            # N > 1: an even number of calls, so that two engines are always in flight and one engine's
            # all-reduce is covered by the other's kernel
            ncalls = max(2, ncalls + (ncalls & 1))
            ncalls = min(ncalls, steps - (steps & 1)) or 2
        b_lo, n_hi = divmod(steps, ncalls)
        b_hi = b_lo + 1 if n_hi else b_lo
        batch = b_hi
        chunks = [b_hi] * (n_hi if n_hi else ncalls) + ([b_lo] * (ncalls - n_hi) if n_hi else [])

        def make(b, seed, stream=None):
            c = pocs_amd.Context(local)
            c.configure(plan, env, K=K, N=N, seed=seed)
            if args.mc_fused:
                c.set_option(pocs_amd.OPT_MC_FUSED, 1)
            if os.environ.get("POCS_NO_STORE") == "1":       # tuning only: samples not written to HBM
                c.set_option(pocs_amd.OPT_STORE_SAMPLES, 0)
            if os.environ.get("POCS_SUB_BATCHES"):           # tuning only: a call's runs as two sub-batches on two streams
                c.set_option(pocs_amd.OPT_SUB_BATCHES, int(os.environ["POCS_SUB_BATCHES"]))
            if os.environ.get("POCS_NO_GRAPH") == "1":       # diagnostic builds that synchronise inside the launch sequence
                c.set_option(pocs_amd.OPT_USE_GRAPH, 0)
            if sharded and not whole:     # one rank per GPU: launches on a torch stream, moments in a torch tensor
                return c, par.GpuEngine(c, W, K, N, rank=rank, world=world, per_rank=n_local, batch=b, stream=stream)
            c.set_batch(b)
            c.set_shard(rank * n_local if whole else 0, n_local)
            if whole:                     # every rank's exchange buffer mapped by every rank (IPC handles over the host channel)
                par.connect_contexts(c, dist if world > 1 else None, rank, world)
            return c, None

        onehop, run_onehop = False, None
        if not sharded or whole:
            # one GPU: each distinct batch size has its own context (and its own captured hipGraph)
            made = [make(b_hi, 0x5EED0001)]
            if b_lo != b_hi and b_lo in chunks:
                made.append(make(b_lo, 0x5EED0002))
            ctx = made[0][0]
            engines = []
            # The launches as they run in production -- the replayed graph -- between ONE pair of hipEvents per call, for
            # every call of the timed region (POCS_OPT_PROFILE = 2): span / W = the mean launch period of the hot kernel,
            # its duration plus the gap to the next launch.  (Events around EVERY launch, the eager form further down,
            # keep the command processor from preparing a launch under the running one: ~9 us more per launch.)
            span = {"ms": 0.0, "n": 0, "on": False}
            if not args.mc_fused and os.environ.get("POCS_NO_GRAPH") != "1" and hasattr(ctx.lib, "pocs_get_exchange_wait"):
                for c, _ in made:
                    c.set_option(pocs_amd.OPT_PROFILE, 2)
                span["on"] = True

            def run_steps(sizes):
                p = 0.0
                for b in sizes:
                    c = made[0][0] if b == b_hi else made[1][0]
                    p = c.run_gmm_estimation() if path == "gmm" else c.run_simulation()
                    if span["on"] and b == b_hi:
                        ms, n = c.kernel_time()
                        span["ms"] += ms
                        span["n"] += n
                return p
        else:
            # one rank per GPU: two engines on two streams take the calls in turn, so one engine's
            # kernel runs while the other's moments are in the all-reduce (parallel.run_gmm_pipelined)
            n_eng = 2 if (path == "gmm" and len(chunks) >= 2 and not fused_exchange) else 1
            if fused_exchange and path == "gmm" and len(chunks) >= 2 and os.environ.get("POCS_ENGINES") == "2":
                # skew tolerance: a second batch of runs in flight on a stream of its own -- while one
                # engine's closers wait for the slowest rank's moments, the other engine's sampling blocks have the chip
                n_eng = 2
            # (every engine on a torch stream of its own: its launches, its collectives and its event waits in one order)
            made = [make(b_hi, 0x5EED0001 + i, torch.cuda.Stream()) for i in range(n_eng)]
            ctx = made[0][0]
            engines = [e for _, e in made]
            # POCS_ONEHOP=1: the library's own exchange (IPC-mapped slots, one hop over xGMI, sum + mixture
            # advance in one small launch) instead of one RCCL all-reduce per waypoint from Python
            onehop = path == "gmm" and xmode in ("1", "3")
            run_onehop = par.run_gmm_onehop_fused if xmode == "3" else par.run_gmm_onehop
            if onehop:
                for e in engines:
                    e.connect_onehop(dist if world > 1 else None, rank, world)

            def run_steps(sizes):
                p = 0.0
                if path != "gmm":
                    for _ in sizes:
                        p = par.run_mc_sharded(engines[0], N, dist)      # one all_reduce of the hit count
                    return p
                for i in range(0, len(sizes), n_eng):
                    group = sizes[i:i + n_eng]
                    for e, b in zip(engines, group):
                        if e.batch != b:
                            e.set_batch(b)
                    p = (run_onehop(engines[:len(group)]) if onehop else par.run_gmm_pipelined(engines[:len(group)], dist))[0]
                return p

        warm = [b_hi] * max(1, (warmup + batch - 1) // batch) + ([b_lo] if b_lo != b_hi else [])          # >= W untimed steps
        if sharded and path == "gmm" and xmode in ("1", "2", "3"):
            # the one-hop exchange has passed a small probe; the warm-up is its first run at full size.  Should it fail
            # there on any rank (the kernel's bounded wait gives up after 30 s and the call returns POCS_E_DEVICE), every
            # rank switches to the RCCL path for the timed region -- agreed by a collective, so that nobody is left behind.
            ok = 1.0
            try:
                run_steps(warm)
            except pocs_amd.PocsError as exc:
                print("rank %d: one-hop exchange failed in the warm-up: %s" % (rank, exc), file=sys.stderr)
                ok = 0.0
            ok = over_ranks([ok], dist.ReduceOp.MIN if dist is not None else None)[0]
            if not ok:
                xstate["mode"], xstate["note"] = "0", "one-hop exchange failed in the warm-up at full size: fell back to RCCL"
                if whole:                                  # (no engines in this form: set the RCCL path up from scratch)
                    fence()
                    for c, _ in made:
                        c.close()
                    return measure(workload, scaling, steps, warmup, batch_arg, samples, full)
                xmode, onehop = "0", False
                run_steps(warm)
        else:
            run_steps(warm)

        # The timed region, repeated: every repeat = EXACTLY `steps` steps between two fences (barrier + synchronize on
        # both sides), its time the MAX over ranks.  The first repeat sizes the rest (the same number on every rank: it
        # derives from the MAX): >= 10 repeats and about TIMED_TARGET_S of GPU time, at most 200.  The line carries
        # the MEDIAN repeat ("median of >= 10 runs after 2 warm-ups") with min / max beside it.
        def timed_pass():
            fence()
            t0 = time.perf_counter()
            p = run_steps(chunks)
            fence()
            d = time.perf_counter() - t0
            if dist is not None:
                d = over_ranks([d], dist.ReduceOp.MAX)[0]
            return d, p
        plain = not sharded or whole
        if plain:
            span["ms"], span["n"] = 0.0, 0             # (the warm-up's spans are not the timed region's)
        d0, prob = timed_pass()                        # (`probability` = the last run of the FIRST repeat: reproducible)
        repeats = int(min(200, max(10, math.ceil(TIMED_TARGET_S / max(d0, 1e-6))))) if full else 3
        dts = [d0] + [timed_pass()[0] for _ in range(repeats - 1)]
        dt = statistics.median(dts)
        evals = float(N) * W * steps
        res = {"workload": workload, "W": W, "K": K, "path": path, "n_local": n_local, "N": N, "batch": batch, "chunks": chunks,
               "prob": prob, "value": evals / dt, "ms_per_step": dt / steps * 1e3, "engines": len(engines) if engines else 1,
               "xmode": xmode, "plan": plan,
               "timing": {"repeats": repeats, "ms_per_step_median": dt / steps * 1e3, "ms_per_step_min": min(dts) / steps * 1e3,
                          "ms_per_step_max": max(dts) / steps * 1e3, "timed_region_s": sum(dts),
                          "protocol": "each repeat = `steps` steps between barrier + synchronize on both sides, MAX over ranks; "
                                      "value and ms_per_step are the median repeat's"}}

        # roofline of the dominant kernel: further calls with the hot kernel bracketed by hipEvents on
        # the launch stream (eager launches; not part of `value`)
        if plain:
            span["on"] = False                             # (the totals of the timed region stay)
        ctx.set_option(pocs_amd.OPT_PROFILE, 1)
        ms_tot, n_launch, groups, waypoint_us = 0.0, 0, 1, None
        for _ in range(max(1, min(len(chunks), 3))):
            if engines:
                if engines[0].batch != b_hi:
                    engines[0].set_batch(b_hi)
                if path == "gmm":
                    (run_onehop(engines[:1]) if onehop else par.run_gmm_pipelined(engines[:1], dist))
                else:
                    par.run_mc_sharded(engines[0], N, dist)
            else:
                run_steps([b_hi])
            ms, n = ctx.kernel_time()
            ms_tot += ms
            n_launch += n
            if path == "gmm" and not engines and hasattr(ctx.lib, "pocs_get_sequence_time"):      # (an A/B library of an older commit has none)
                seq_ms, groups = ctx.sequence_time()
                waypoint_us = seq_ms * 1e3 / W
        res["bracketed_ms"] = ms_tot / max(n_launch, 1)            # events around every launch (eager)
        res["avg_ms"], res["duration_is"] = res["bracketed_ms"], "hipEvents around every launch of the hot kernel (eager launches), mean"
        res["span_timed"] = plain and span["n"] > 0
        if res["span_timed"]:
            res["avg_ms"] = span["ms"] / span["n"]
            res["duration_is"] = ("mean PERIOD of a waypoint inside the replayed graph over every call of the timed region: one pair of hipEvents "
                                  "around each graph launch, span / waypoints -- with one launch per waypoint the kernel's duration plus the gap to "
                                  "the next launch (an upper bound of the duration); with `concurrent_launches` = 2 the call's runs go out as two "
                                  "launches of half the runs each on two streams, side by side, and the period is that of BOTH: achieved = "
                                  "algorithmic_bytes_per_period / period, the chip's rate (a profiler's kernel trace shows the half-size launches, each "
                                  "lasting about a period, and keeps them from overlapping as they do here); `bracketed_kernel_us` = events around "
                                  "every single launch of sub-batch 0 instead")
        res["groups"], res["waypoint_us"] = groups, waypoint_us
        # sharded through the library's exchange: how long this rank's closers waited for the other ranks' moments in
        # that last call, per (run, waypoint)
        res["xwait"] = None
        if ((engines and onehop) or whole) and hasattr(ctx.lib, "pocs_get_exchange_wait"):
            try:
                res["xwait"] = ctx.exchange_wait_us()
            except pocs_amd.PocsError:
                res["xwait"] = None
        # the same launches WITHOUT the sample stores (POCS_OPT_STORE_SAMPLES = 0: a product option, same arithmetic, same
        # results): what the arithmetic alone takes on this box, to set beside what the stream alone would take at the fill
        # rate -- the two meet at the board's power cap; not part of `value`
        res["nostore_ms"] = None
        if full and path == "gmm" and not engines and not whole and os.environ.get("POCS_NO_STORE") != "1":
            if res["span_timed"]:                                # timed the same way as the launches it is set beside
                ctx.set_option(pocs_amd.OPT_PROFILE, 2)
            ctx.set_option(pocs_amd.OPT_STORE_SAMPLES, 0)
            run_steps([b_hi])                            # (the graph / buffers of this variant)
            t_ms, t_n = 0.0, 0
            for _ in range(max(1, min(len(chunks), 2))):
                run_steps([b_hi])
                ms, n = ctx.kernel_time()
                t_ms += ms
                t_n += n
            ctx.set_option(pocs_amd.OPT_STORE_SAMPLES, 1)
            res["nostore_ms"] = t_ms / max(t_n, 1)
        ctx.set_option(pocs_amd.OPT_PROFILE, 0)
        # shader clock under this load, in calls of their own: a sensor read goes through the driver and
        # disturbs the GPU (kernels 15 % slower while it polls), so nothing else is measured meanwhile
        res["board"] = None
        if full and rank == 0 and not sharded and os.environ.get("POCS_NO_BOARD_PROBE") != "1":
            try:
                bus = getattr(torch.cuda.get_device_properties(local), "pci_bus_id", None)
                probe = BoardProbe("%02x:00.0" % bus if isinstance(bus, int) else None)
                probe.start()
                t_end = time.perf_counter() + 0.4
                while time.perf_counter() < t_end:
                    run_steps([b_hi])
                res["board"] = probe.stop()
            except Exception:                            # never let the side measurement break the bench line
                res["board"] = None
        res["copy_gbps"] = ctx.copy_bandwidth(1 << 30) if full else None      # measured streaming-copy ceiling of this GPU, same process
        res["fill_gbps"] = ctx.fill_bandwidth(1 << 30) if full else None      # ... and the write-only one (the GMM kernels read nothing)
        res["ranks_kernel_us"] = gather_ranks(res["avg_ms"] * 1e3) if dist is not None else None

        # one run per call, no batch, no run-ahead: the rate of ONE runGMMEstimation / runSimulation command
        # (SURVEY 8d defines the metric on one run* call; `value` above is the batched throughput)
        res["single"] = None
        if full and not sharded and os.environ.get("POCS_SKIP_SINGLE") != "1":
            with pocs_amd.Context(local) as c1:
                c1.configure(plan, env, K=K, N=N, seed=0x5EED0003)
                c1.set_shard(0, n_local)
                if args.mc_fused:
                    c1.set_option(pocs_amd.OPT_MC_FUSED, 1)
                run1 = c1.run_gmm_estimation if path == "gmm" else c1.run_simulation
                for _ in range(3):
                    run1()
                torch.cuda.synchronize()
                reps = 12
                t1 = time.perf_counter()
                for _ in range(reps):
                    run1()
                torch.cuda.synchronize()
                res["single"] = float(n_local) * W * reps / (time.perf_counter() - t1)
        if dist is not None:
            fence()                                        # every rank is done with every rank's exchange buffer
        for c, _ in made:
            c.close()
        return res

    def exchange_text(res):
        if not sharded or res["path"] != "gmm":
            return "none (one GPU)" if not sharded else "none on the data path (MC: one all-reduce of the hit counts per run)"
        return ("one-hop IPC slots in the sampling launches' tails, the whole call replayed from a graph by the library (a connected context's pocs_run_gmm_estimation)" if res["xmode"] == "2"
                else "one-hop IPC slots in the sampling launch's tail, one launch per waypoint issued from the host (pocs_gmm_sample_exchange_local)" if res["xmode"] == "3"
                else "one-hop IPC slots (pocs_gmm_exchange_local)" if res["xmode"] == "1"
                else ("RCCL all-reduce per waypoint" + (" (%s)" % xstate["note"] if xstate["note"] else "")))

    def exchange_wait(res):
        """The closers' waits for the other ranks' moments: every rank's (min, median, max) over the (run, waypoint) pairs of
        its last profiled call; over ranks the min of the mins, min / median / max of the medians, the max of the maxes."""
        if dist is None or res["path"] != "gmm":
            return None
        mine = res["xwait"] if res["xwait"] is not None else (float("nan"),) * 3
        cols = [gather_ranks(v) for v in mine]
        if any(v != v for v in cols[1]):
            return {"available": False, "why": "no in-kernel exchange on this path (%s)" % exchange_text(res)}
        return {"available": True, "min_us": min(cols[0]), "median_us": {"min": min(cols[1]), "median": statistics.median(cols[1]), "max": max(cols[1]),
                                                                        "per_rank": cols[1]}, "max_us": max(cols[2]),
                "what": "in-kernel wall clock around the closer's poll for the world's rows, per (run, waypoint) of one call "
                        "(pocs_get_exchange_wait); a rank that runs ahead of the others waits here",
                "status": "measured on this run's ranks" + (" -- ALL ON ONE CARD (rehearsal): unmeasured on hardware with more than one GPU" if "POCS_FORCE_DEVICE" in os.environ or world == 1 else "")}

    main_res = measure(args.workload, args.scaling, args.steps, args.warmup, args.batch, args.samples, True)
    strong_rec = None
    if (world > 1 and args.scaling == "weak" and args.workload == "cfg2" and not args.no_strong_record and not args.mc_fused):
        # north_star's >= 6 x is about BASELINE configs[3] (cfg3's 10^7 samples x 500 waypoints, K = 8, SPLIT over the
        # GPUs): the default curve above is weak scaling on cfg2, so a short pass of that workload rides along
        # (64 runs per call: a rank's launch then holds 64 x 10^7 / N evaluations -- at N = 8 what a 64-run launch of the
        # default workload holds -- so that the launch's fixed cost and the exchange weigh as they do there)
        # (the one-card rehearsal keeps 16: two ranks' 512-block launches do not fit ONE card side by side, and a closer
        # that waits for a rank whose launch cannot start waits for its 30 s)
        sruns = int(os.environ.get("POCS_STRONG_RUNS", "16" if "POCS_FORCE_DEVICE" in os.environ else "64"))
        sres = measure("cfg3", "strong", sruns, sruns, sruns, args.samples, False)
        kus = sres["ranks_kernel_us"] or []
        strong_rec = {"workload": "cfg3 split over %d GPUs (BASELINE configs[3] at 8): %d samples per GPU per run of %d in total, W=%d, K=%d, %d runs per call"
                                  % (world, sres["n_local"], sres["N"], sres["W"], sres["K"], sres["batch"]),
                      "scaling": "strong", "value": sres["value"], "ms_per_step": sres["ms_per_step"], "repeats": sres["timing"]["repeats"],
                      "ms_per_step_min": sres["timing"]["ms_per_step_min"], "ms_per_step_max": sres["timing"]["ms_per_step_max"],
                      "probability": sres["prob"], "exchange": exchange_text(sres), "exchange_wait_us": exchange_wait(sres),
                      "ranks_kernel_us": {"min": min(kus), "max": max(kus), "all": kus} if kus else None,
                      "skew_us": (max(kus) - min(kus)) if kus else None,
                      "one_gpu_reference": "the same workload on one GPU: `bench.py --workload cfg3 --steps 16 --warmup 16` (1.66-1.70e11 evals/s whatever the runs per call: DESIGN.md section 7)"}

    res = main_res
    W, K, path, n_local, N, batch, chunks = res["W"], res["K"], res["path"], res["n_local"], res["N"], res["batch"], res["chunks"]
    board, nostore_ms, groups, avg_ms = res["board"], res["nostore_ms"], res["groups"], res["avg_ms"]
    copy_gbps, fill_gbps = res["copy_gbps"], res["fill_gbps"]
    kern = "k_gmm_step" if path == "gmm" else ("k_mc_fused" if args.mc_fused else "k_mc_step")
    bpe = BYTES_PER_EVAL_GMM if path == "gmm" else BYTES_PER_EVAL_MC
    # a call may be issued as G sub-batches whose launches run side by side (POCS_OPT_SUB_BATCHES, default 1): the events
    # bracket sub-batch 0's launches, each of which works on batch / G runs
    # A call of many runs goes out as G sub-batches on G streams (POCS_OPT_SUB_BATCHES; two by default where the call has
    # the work for it): G launches of batch / G runs run SIDE BY SIDE, and the span-timed period is that of a waypoint of
    # the whole call -- so the rate is the chip's: the bytes of all G launches over the period.  (Bracketed timing, the
    # fallback, times sub-batch 0's launches alone: one launch's bytes over its duration.)
    per_launch = n_local * batch // max(groups, 1)  # evaluations one launch of the hot kernel processes
    units = n_local * batch if res.get("span_timed") else per_launch       # ... and one period
    achieved = (bpe * units) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # Committed counter records of this workload (separate rocprofv3 --pmc passes, gfx950 FETCH_SIZE correction applied;
    # profiles/traffic.json with its sources): HBM bytes and vector instructions per launch.  The record nearest in launch
    # size is scaled to THIS launch's evaluations -- an extrapolation, said so in `traffic_source`, and dropped (null)
    # when the nearest record's launch is more than 2 x away.
    traffic, traffic_src, valu_per_eval, valu_src = None, None, None, None
    tj = ROOT / "profiles" / "traffic.json"
    if tj.exists():
        recs = [(k, v) for k, v in json.loads(tj.read_text()).items()
                if v.get("path", "gmm" if k.startswith("cfg") and not k.startswith("cfg5") else "mc") == path and v.get("evals_per_launch")
                and v.get("numerics", "v6") == NUMERICS
                and (path != "gmm" or (v.get("components") or (8 if "cfg3" in k else 3)) == K)]     # another K is another kernel
        if recs:
            k, rec = min(recs, key=lambda kv: abs(kv[1]["evals_per_launch"] - units))
            if 0.5 <= rec["evals_per_launch"] / units <= 2.0:
                traffic = rec["bytes_per_launch"] / rec["evals_per_launch"] * units
                traffic_src = "%.3f B/eval measured at %d evals per launch (%s), scaled to %d evals" % (
                    rec["bytes_per_launch"] / rec["evals_per_launch"], rec["evals_per_launch"], rec["source"].split(":")[0], units)
                if rec.get("valu_insts_per_launch"):
                    valu_per_eval = rec["valu_insts_per_launch"] / (rec["evals_per_launch"] / 64.0)   # wave instructions per wave's 64 evaluations
                    valu_src = rec["source"].split(":")[0]
    # What the kernel is actually limited by, in numbers: vector instructions per evaluation (counter record above) x
    # the evaluations per second inside the kernel, against the chip's wave64 issue rate -- 256 CUs x 4 SIMDs, one vector
    # instruction per SIMD every 4 cycles -- at the shader clock read live under this load (`board`), or at the peak clock.
    mc_state_bytes = float(batch) * float((n_local + 1) & ~1) * 28.0
    mc_resident = path == "mc" and not args.mc_fused and mc_state_bytes <= MC_CACHE_BYTES
    if path == "gmm":
        clock = (board or {}).get("sclk_MHz") or 2400.0
        issue_peak = 1024 * clock * 1e6 / 4.0
        bound = "fp64-issue+power"
        limiter = {"kind": "FP64 vector issue (+ per-lane LDS table reads) and the all-write sample stream, coupled through the "
                           "board's power cap: neither alone (DESIGN.md section 5)",
                   "kernel_us_without_sample_stores": nostore_ms * 1e3 if nostore_ms else None,
                   "stream_alone_us_at_fill_rate": (bpe * units / (fill_gbps * 1e9) * 1e6) if (fill_gbps or 0) > 0 else None,
                   "valu_instr_per_eval": valu_per_eval, "clock_MHz": clock,
                   "clock_source": "amdgpu hwmon, live" if (board or {}).get("sclk_MHz") else "peak clock (no live reading)",
                   "valu_issue_frac": (valu_per_eval * (units / 64.0) / (avg_ms * 1e-3) / issue_peak) if (valu_per_eval and avg_ms > 0) else None,
                   "valu_source": valu_src,
                   "power_W": (board or {}).get("power_W"), "power_cap_W": (board or {}).get("power_cap_W")}
    elif args.mc_fused:
        bound = "fp64-issue"
        limiter = {"kind": "FP64 vector issue: the particle stays in registers for the whole roll-out"}
    elif mc_resident:
        bound = "infinity-cache"
        limiter = {"kind": "streaming through the 256 MB Infinity Cache: the batch's particle state (%.0f MB = %d runs x %d particles x 28 B) "
                           "stays resident between two waypoint launches BY DESIGN (bench.py sizes the batch for it, pocs_host.hip picks the "
                           "plain-access kernel), so the bytes per second below are not HBM traffic; the HBM-streaming form of the same kernel "
                           "(k_mc_step<NT>) is what `--batch` beyond the cache measures (profiles/r04_mc_nt_*)" % (mc_state_bytes / 1e6, batch, n_local)}
    else:
        bound = "hbm"
        limiter = {"kind": "HBM streaming (particle state read and written per waypoint, non-temporal: %.0f MB of state per launch do not fit "
                           "the 256 MB Infinity Cache; the hit counter only where a particle collides)" % (mc_state_bytes / 1e6)}
    # `frac` is quoted against the HBM peak wherever the bytes can be HBM bytes: not for the cache-resident MC batch (null
    # there: the guide gives no peak for the Infinity Cache) -- its achieved rate stands beside the copy ceiling instead
    hbm_frac = None if mc_resident else achieved / HBM_PEAK_GBPS
    roofline = {"bound": bound, "reported_against": None if mc_resident else "hbm", "limiter": limiter, "board": board,
                "kernel": kern, "achieved": achieved, "peak": None if mc_resident else HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": hbm_frac, "traffic": traffic, "traffic_source": traffic_src,
                "traffic_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if (traffic and avg_ms > 0 and not mc_resident) else None,
                "copy_GBps": copy_gbps, "frac_of_copy": achieved / copy_gbps if (copy_gbps or 0) > 0 else None,
                "fill_GBps": fill_gbps, "frac_of_fill": achieved / fill_gbps if (fill_gbps or 0) > 0 else None,
                "algorithmic_bytes_per_launch": bpe * per_launch, "bytes_per_eval": bpe, "evals_per_launch": per_launch,
                "concurrent_launches": groups, "evals_per_period": units, "algorithmic_bytes_per_period": bpe * units,
                "avg_kernel_us": avg_ms * 1e3, "duration_is": res["duration_is"], "bracketed_kernel_us": res["bracketed_ms"] * 1e3,
                "waypoint_us": res["waypoint_us"],
                "evals_per_s_in_kernel": units / (avg_ms * 1e-3) if avg_ms > 0 else 0.0}
    if path == "mc" and not args.mc_fused:
        roofline["resident"] = "infinity-cache" if mc_resident else "hbm"
        roofline["state_bytes_per_launch"] = mc_state_bytes
    if args.mc_fused:
        roofline["note"] = ("fused roll-out: the particle stays in registers for all waypoints, ~0 algorithmic bytes per evaluation "
                            "(28 B per particle per RUN) -- an FP64-issue kernel whose roofline is not HBM; read evals_per_s_in_kernel, "
                            "not frac (SURVEY 8d: reported separately from the streaming kernel)")
    if dist is not None:                               # every rank's kernel time, so that a scaling run explains itself
        kus = res["ranks_kernel_us"]
        roofline["ranks_kernel_us"] = {"min": min(kus), "max": max(kus), "all": kus}
        roofline["skew_us"] = max(kus) - min(kus)      # slowest - fastest rank's mean launch of the hot kernel

    t_gpu_part = time.perf_counter() - t_process
    if rank == 0:
        out = {
            "metric": "particle-waypoint evals/s (GMM+collision)" if path == "gmm" else "particle-waypoint evals/s (MC+collision)",
            "value": res["value"], "unit": "particle-waypoint evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "numerics": "%s (%s)" % (NUMERICS, pocs_amd.load_library().pocs_version().decode()),
            "timing": res["timing"],
            "single_call_evals_per_s": res["single"],
            "config": {"workload": "%s: %s path, %s plan (%d waypoints), %d samples per GPU per run, K=%d, pr2test2 walls, PR2 0.668 m square footprint"
                                   % (args.workload, path.upper(), "bundled trajectory.dat/odometry.dat" if W == 56 else "resampled", W, n_local, K),
                       "waypoints": W, "samples_per_gpu": n_local, "components": K, "probability": res["prob"],
                       "runs_per_launch": batch, "calls": chunks,
                       "engines_in_flight": res["engines"],
                       "exchange": exchange_text(res),
                       "total_samples_per_run": N,
                       "random_stream": "Philox4x32-7 for the mixture samples (SURVEY 8d wrote Philox4x32-10: a stated deviation, DESIGN.md section 4), "
                                        "Philox4x32-10 for the host chain, the initial particles and the component counts",
                       "value_is": "batched throughput: `runs_per_launch` independent runs (the reference driver's 200-run loop) "
                                   "advance in lockstep per call; single_call_evals_per_s = one run per call",
                       "sanity_band": "collision model = this build's 2-D boxes, not OpenRAVE/ODE + PR2 mesh (not in the reference tree): "
                                      "200 runs at N = 10^4 give MC 0.706 / GMM 0.285 (profiles/r01_table1_like.txt) against the paper's "
                                      "0.935 / 0.64; same ordering, different level -- parity unpinned at that call site (DESIGN.md 8)"},
            "roofline": roofline,
        }
        if sharded:
            out["exchange_probe"] = xstate["probe"]
            out["exchange_wait_us"] = None
    if sharded:                                            # (collectives: every rank takes part)
        xw = exchange_wait(res)
        if rank == 0:
            out["exchange_wait_us"] = xw
            if strong_rec is not None:
                out["strong"] = strong_rec
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(res["plan"], env, K, W, path, args.cpu_evals, budget_s=max(2.0, CPU_BUDGET_FACTOR * t_gpu_part))
        sys.stdout.flush()
        os.dup2(json_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
