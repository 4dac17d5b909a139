#!/usr/bin/env python3
"""bench.py -- particle-waypoint evaluations/s of the GMM + collision hot path on MI355X.

One "step" = one runGMMEstimation over the whole plan (W waypoints x N samples) on synthetic
(seeded) draws.  Default workload at --gpus 1 is BASELINE.json configs[1]: the bundled
trajectory.dat / odometry.dat plan (56 waypoints), 10^6 samples, 3-component mixture.
With --gpus G (launched by torch.distributed.run, one rank per GPU) every rank evaluates its
own 10^6 samples of a G x 10^6 mixture (weak scaling); the per-waypoint moments (11 K doubles)
are summed over ranks with one RCCL all-reduce per waypoint.

Prints ONE JSON line on rank 0 (see DESIGN.md section 7 for every field).
"""
import argparse
import json
import os
import sys
import time
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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--samples", type=int, default=0, help="override samples per GPU (experiments only)")
    ap.add_argument("--concurrent", type=int, default=1,
                    help="independent runs kept in flight per GPU (one context + stream each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-evals", type=float, default=6.0e7, help="size of the CPU baseline sample")
    args = ap.parse_args()

    import torch
    import pocs_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist = None
    sharded = world > 1 or os.environ.get("POCS_FORCE_SHARDED") == "1"   # rehearse the N>1 path on one GPU
    if sharded:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(local)

    W, n_local, K, path = WORKLOADS[args.workload]
    if args.samples:
        n_local = args.samples
    plan = pocs_amd.load_plan()
    if W != 56:
        plan = pocs_amd.resample_plan(plan, W)
    env = pocs_amd.load_env()
    N = n_local * world

    from importlib import import_module
    par = import_module("probability-of-collision-for-safe-planning_amd.parallel")
    ctx = pocs_amd.Context(local)
    ctx.configure(plan, env, K=K, N=N, seed=0x5EED0001)
    engine = None
    if sharded:
        # one rank per GPU: launches on torch's stream, moments in a torch tensor for all_reduce
        engine = par.GpuEngine(ctx, W, K, N, rank=rank, world=world, per_rank=n_local)
    else:
        ctx.set_shard(0, n_local)

    def step():
        if path == "gmm":
            if not sharded:
                return ctx.run_gmm_estimation()          # whole run replayed from one hipGraph
            return par.run_gmm_sharded(engine, dist)     # per waypoint: step_local + all_reduce(11K f64)
        if not sharded:
            return ctx.run_simulation()
        return par.run_mc_sharded(engine, N, dist)       # one all_reduce of the hit count

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # optional: R independent runs in flight (the reference's driver performs 200 independent
    # runs, MCSimulation.py:238-256); each has its own context, stream and buffers, and a host
    # thread that issues its share of the K steps.  Single-GPU only.
    extra = []
    if args.concurrent > 1 and world == 1:
        import threading
        for r in range(1, args.concurrent):
            c2 = pocs_amd.Context(local)
            c2.configure(plan, env, K=K, N=N, seed=0x5EED0001 + 977 * r)
            c2.set_shard(0, n_local)
            extra.append(c2)
        ctxs = [ctx] + extra

        def run_many(total):
            def work(c, n):
                for _ in range(n):
                    c.run_gmm_estimation() if path == "gmm" else c.run_simulation()
            share = [total // len(ctxs) + (1 if i < total % len(ctxs) else 0) for i in range(len(ctxs))]
            th = [threading.Thread(target=work, args=(c, n)) for c, n in zip(ctxs, share) if n]
            for t in th:
                t.start()
            for t in th:
                t.join()

    for _ in range(args.warmup):
        step()
    if extra:
        run_many(len(ctxs) * 2)
    fence()
    t0 = time.perf_counter()
    prob = 0.0
    if extra:
        run_many(args.steps)
        prob = step() if False else 0.0
    else:
        for _ in range(args.steps):
            prob = step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    evals = float(N) * W * args.steps
    value = evals / dt

    # roofline of the dominant kernel: a second pass of the same steps with the hot kernel
    # bracketed by hipEvents on the launch stream (eager launches; not part of `value`)
    ctx.set_option(pocs_amd.OPT_PROFILE, 1)
    ms_tot, n_launch = 0.0, 0
    for _ in range(max(1, min(args.steps, 5))):
        step()
        ms, n = ctx.kernel_time()
        ms_tot += ms
        n_launch += n
    ctx.set_option(pocs_amd.OPT_PROFILE, 0)
    kern = "k_gmm_step" if path == "gmm" else "k_mc_step"
    bpe = BYTES_PER_EVAL_GMM if path == "gmm" else BYTES_PER_EVAL_MC
    avg_ms = ms_tot / max(n_launch, 1)
    achieved = (bpe * n_local) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc passes, gfx950 FETCH_SIZE
    # correction applied) are taken offline and committed with their source in profiles/traffic.json
    traffic, traffic_src = None, None
    tj = ROOT / "profiles" / "traffic.json"
    if tj.exists() and not args.samples:
        rec = json.loads(tj.read_text()).get(args.workload)
        if rec:
            traffic, traffic_src = rec["bytes_per_launch"], rec["source"]
    roofline = {"bound": "hbm", "kernel": kern, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bpe * n_local,
                "bytes_per_eval": bpe, "evals_per_launch": n_local, "avg_kernel_us": avg_ms * 1e3,
                "evals_per_s_in_kernel": n_local / (avg_ms * 1e-3) if avg_ms > 0 else 0.0}

    if rank == 0:
        out = {
            "metric": "particle-waypoint evals/s (GMM+collision)" if path == "gmm" else "particle-waypoint evals/s (MC+collision)",
            "value": value, "unit": "particle-waypoint evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s path, %s plan (%d waypoints), %d samples per GPU, K=%d, pr2test2 walls, PR2 0.668 m square footprint"
                                   % (args.workload, path.upper(), "bundled trajectory.dat/odometry.dat" if W == 56 else "resampled", W, n_local, K),
                       "waypoints": W, "samples_per_gpu": n_local, "components": K, "probability": prob,
                       "concurrent_runs": args.concurrent if world == 1 else 1},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(plan, env, K, W, path, args.cpu_evals)
        print(json.dumps(out))
    for c2 in extra:
        c2.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
