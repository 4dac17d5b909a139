#!/usr/bin/env python3
"""Tuning aid: ways of keeping one GPU busy across the per-waypoint all-reduce of the sharded GMM
path, compared in one process (RCCL, world size 1) through parallel.run_gmm_pipelined: one engine on
one stream; two engines on two streams (bench.py's choice for N > 1); two engines sharing a stream;
and the unsharded call for scale.  (Also tried: the exchange and the mixture advance on a side
stream per engine -- every cross-stream hop costs, 4-5 % slower than either.)"""
import os, sys, time
from importlib import import_module
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import pocs_amd
par = import_module("probability-of-collision-for-safe-planning_amd.parallel")

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
N, K, W = 1000000, 3, 56
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 64


def make(n_eng, shared):
    compute = torch.cuda.Stream()
    ctxs, engs = [], []
    for i in range(n_eng):
        c = pocs_amd.Context(0); c.configure(plan, env, K=K, N=N, seed=11 + i); ctxs.append(c)
        engs.append(par.GpuEngine(c, W, K, N, rank=0, world=1, per_rank=N, batch=batch,
                                  stream=compute if shared else torch.cuda.Stream()))
    return ctxs, engs


for rnd in range(2):
    for name, n_eng, shared, fn in (("1 engine, one stream", 1, True, par.run_gmm_pipelined),
                                    ("2 engines, two streams", 2, False, par.run_gmm_pipelined),
                                    ("2 engines, one stream", 2, True, par.run_gmm_pipelined)):
        ctxs, engs = make(n_eng, shared)
        fn(engs, dist); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            fn(engs, dist)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%-48s %.3e evals/s" % (name, 2 * n_eng * batch * N * W / dt), flush=True)
        for c in ctxs: c.close()
    with pocs_amd.Context(0) as c:
        c.configure(plan, env, K=K, N=N, seed=3); c.set_batch(batch)
        for g in (1, 0):
            c.set_option(pocs_amd.OPT_USE_GRAPH, g)
            c.run_gmm_estimation()
            t0 = time.perf_counter()
            for _ in range(3): c.run_gmm_estimation()
            dt = time.perf_counter() - t0
            print("%-48s %.3e evals/s" % ("unsharded call, graph %d" % g, 3 * batch * N * W / dt), flush=True)
dist.destroy_process_group()
