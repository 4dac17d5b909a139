#!/usr/bin/env python3
"""bench.py -- particle-waypoint evaluations/s of the GMM + collision hot path on MI355X.

One "step" = one runGMMEstimation over the whole plan (W waypoints x N samples) on synthetic
(seeded) draws -- one iteration of the reference driver's 200-run loop (MCSimulation.py:238-256).
Steps are independent, so the GPU advances `--batch` of them in lockstep per call (one launch per
waypoint for the whole batch, pocs_set_batch); K steps = K/batch calls (+ one call with the
remainder).  Default workload at --gpus 1 is BASELINE.json configs[1]: the bundled
trajectory.dat / odometry.dat plan (56 waypoints), 10^6 samples, 3-component mixture.

With --gpus G (launched by torch.distributed.run, one rank per GPU) every rank evaluates its own
10^6 samples of a G x 10^6-sample mixture (weak scaling); per waypoint the moments of the whole
batch (batch x 11 K doubles) are summed over ranks with ONE RCCL all-reduce.

Prints ONE JSON line on rank 0 (fields: DESIGN.md section 7).
"""
import argparse
import json
import os
import sys
import time
from importlib import import_module
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_EVAL_GMM = 26         # 3 x f64 sample + i16 flag streamed out once
BYTES_PER_EVAL_MC = 56          # 24 B in + 24 B out + u32 hit counter read + write

WORKLOADS = {
    # name: (waypoints, samples per GPU, components, path)
    "cfg2": (56, 1_000_000, 3, "gmm"),      # bundled plan, 1M samples, K=3   <- the metric's config
    "cfg3": (500, 10_000_000, 8, "gmm"),    # 500-waypoint resampled plan, 10M samples, K=8
    "cfg5": (500, 100_000, 1, "mc"),        # MC roll-outs, 500 waypoints
    "mc": (56, 1_000_000, 1, "mc"),
}


def cpu_baseline(plan, env, K, W, path, budget_evals):
    """The oracle (single-thread C restatement) timed on this host on a bounded sample."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import oracle
    orc = oracle.Oracle()
    cfg = orc.config(plan, env, K=K)
    n = max(1000, int(budget_evals // W))
    t0 = time.perf_counter()
    if path == "gmm":
        orc.run_gmm(cfg, 1234, n)
    else:
        orc.run_mc(cfg, 1234, n)
    dt = time.perf_counter() - t0
    return {"value": n * W / dt, "unit": "particle-waypoint evals/s", "cores": 1, "kind": "port",
            "sample": "%d samples x %d waypoints (%s path, K=%d), oracle/pocs_oracle.c, 1 thread, %.1f s"
                      % (n, W, path, K, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=8,
                    help="independent runs (steps) advanced in lockstep per call (pocs_set_batch)")
    ap.add_argument("--samples", type=int, default=0, help="override samples per GPU (experiments only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-evals", type=float, default=6.0e7, help="size of the CPU baseline sample")
    args = ap.parse_args()

    import torch
    import pocs_amd
    par = import_module("probability-of-collision-for-safe-planning_amd.parallel")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dist = None
    sharded = world > 1 or os.environ.get("POCS_FORCE_SHARDED") == "1"   # rehearse the N>1 path on one GPU
    if sharded:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    W, n_local, K, path = WORKLOADS[args.workload]
    if args.samples:
        n_local = args.samples
    plan = pocs_amd.load_plan()
    if W != 56:
        plan = pocs_amd.resample_plan(plan, W)
    env = pocs_amd.load_env()
    N = n_local * world

    batch = max(1, min(args.batch, args.steps)) if path == "gmm" else 1
    rem = args.steps % batch

    def make(b, seed):
        c = pocs_amd.Context(local)
        c.configure(plan, env, K=K, N=N, seed=seed)
        if sharded:     # one rank per GPU: launches on torch's stream, moments in a torch tensor
            return c, par.GpuEngine(c, W, K, N, rank=rank, world=world, per_rank=n_local, batch=b)
        c.set_batch(b)
        c.set_shard(0, n_local)
        return c, None

    ctx, engine = make(batch, 0x5EED0001)
    ctx_rem, engine_rem = make(rem, 0x5EED0002) if rem else (None, None)

    def call(c, e):
        """One call = `batch` steps (GMM) or one step (MC)."""
        if path == "gmm":
            if not sharded:
                return c.run_gmm_estimation()            # the whole batch replayed from one hipGraph
            return par.run_gmm_sharded(e, dist)          # per waypoint: step_local + all_reduce(batch*11K f64)
        if not sharded:
            return c.run_simulation()
        return par.run_mc_sharded(e, N, dist)            # one all_reduce of the hit count

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range((args.warmup + batch - 1) // batch):      # >= W untimed steps
        call(ctx, engine)
    if ctx_rem is not None:
        call(ctx_rem, engine_rem)
    fence()
    t0 = time.perf_counter()
    prob = 0.0
    for _ in range(args.steps // batch):
        prob = call(ctx, engine)
    if ctx_rem is not None:
        prob = call(ctx_rem, engine_rem)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    evals = float(N) * W * args.steps
    value = evals / dt

    # roofline of the dominant kernel: further calls with the hot kernel bracketed by hipEvents on
    # the launch stream (eager launches; not part of `value`)
    ctx.set_option(pocs_amd.OPT_PROFILE, 1)
    ms_tot, n_launch = 0.0, 0
    for _ in range(max(1, min(args.steps // batch, 3))):
        call(ctx, engine)
        ms, n = ctx.kernel_time()
        ms_tot += ms
        n_launch += n
    ctx.set_option(pocs_amd.OPT_PROFILE, 0)
    kern = "k_gmm_step" if path == "gmm" else "k_mc_step"
    bpe = BYTES_PER_EVAL_GMM if path == "gmm" else BYTES_PER_EVAL_MC
    avg_ms = ms_tot / max(n_launch, 1)
    units = n_local * batch                   # evaluations one launch of the hot kernel processes
    achieved = (bpe * units) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes, gfx950 FETCH_SIZE
    # correction applied) are taken offline and committed with their source in profiles/traffic.json
    traffic, traffic_src = None, None
    tj = ROOT / "profiles" / "traffic.json"
    if tj.exists() and not args.samples:
        rec = json.loads(tj.read_text()).get("%s_batch%d" % (args.workload, batch))
        if rec:
            traffic, traffic_src = rec["bytes_per_launch"], rec["source"]
    roofline = {"bound": "hbm", "kernel": kern, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bpe * units, "bytes_per_eval": bpe, "evals_per_launch": units,
                "avg_kernel_us": avg_ms * 1e3,
                "evals_per_s_in_kernel": units / (avg_ms * 1e-3) if avg_ms > 0 else 0.0}

    if rank == 0:
        out = {
            "metric": "particle-waypoint evals/s (GMM+collision)" if path == "gmm" else "particle-waypoint evals/s (MC+collision)",
            "value": value, "unit": "particle-waypoint evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s path, %s plan (%d waypoints), %d samples per GPU per run, K=%d, pr2test2 walls, PR2 0.668 m square footprint"
                                   % (args.workload, path.upper(), "bundled trajectory.dat/odometry.dat" if W == 56 else "resampled", W, n_local, K),
                       "waypoints": W, "samples_per_gpu": n_local, "components": K, "probability": prob,
                       "runs_per_launch": batch},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(plan, env, K, W, path, args.cpu_evals)
        print(json.dumps(out))
    for c in (ctx, ctx_rem):
        if c is not None:
            c.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
