"""The N > 1 GPU path end to end on ONE card: two processes share the GPU (gloo moves the moments
between them), each evaluates its shard through libpocs.so with parallel.GpuEngine /
run_gmm_pipelined / run_mc_sharded -- exactly what bench.py does per rank over RCCL.  The sharded
result must equal the one-process result for the same total sample count."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, n_local, K, seed, batch, out):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from importlib import import_module
    import pocs_amd
    par = import_module("probability-of-collision-for-safe-planning_amd.parallel")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
    N = n_local * world
    ctxs, engs = [], []
    for i in range(2):                                    # two engines in flight, as in bench.py
        c = pocs_amd.Context(0)
        c.configure(plan, env, K=K, N=N, seed=seed + i)
        ctxs.append(c)
        engs.append(par.GpuEngine(c, 56, K, N, rank=rank, world=world, per_rank=n_local, batch=batch,
                                  stream=torch.cuda.Stream()))
    par.run_gmm_pipelined(engs, dist)
    torch.cuda.synchronize()
    probs = [list(e.probabilities()) for e in engs]
    ctxs[0].set_seed(seed)
    p_mc = par.run_mc_sharded(engs[0], N, dist)
    mc_all = engs[0].last_mc_probabilities
    if rank == 0:
        np.save(out, np.array(probs[0] + probs[1] + mc_all + [p_mc]))
    for c in ctxs:
        c.close()
    dist.destroy_process_group()


def _worker_onehop(rank, world, port, n_local, K, seed, batch, out):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from importlib import import_module
    import pocs_amd
    par = import_module("probability-of-collision-for-safe-planning_amd.parallel")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
    N = n_local * world
    res = {}
    for mode in ("gloo", "onehop", "onehop_again", "fused"):
        c = pocs_amd.Context(0)
        c.configure(plan, env, K=K, N=N, seed=seed)
        e = par.GpuEngine(c, 56, K, N, rank=rank, world=world, per_rank=n_local, batch=batch, stream=torch.cuda.Stream())
        if mode == "gloo":
            par.run_gmm_pipelined([e], dist)
        elif mode == "fused":                             # sample + exchange + advance in one launch per waypoint
            e.connect_onehop(dist, rank, world)
            par.run_gmm_onehop_fused([e])
        else:
            e.connect_onehop(dist, rank, world)
            par.run_gmm_onehop([e])
            if mode == "onehop_again":                    # a second call on the same buffers: epochs move on
                par.run_gmm_onehop([e])
        torch.cuda.synchronize()
        res[mode] = (list(e.probabilities()), np.array([c.moments(w, K) for w in range(56)]))
        dist.barrier()
        c.close()
    # the WHOLE CALL in one library call (what bench.py --gpus N uses since round 4): a context with its shard set, connected
    # to its peers, replays the call from a graph and exchanges every run's moments in the closing block of its launch --
    # as one launch per waypoint, as two sub-batches on two streams, and a second call on the same buffers (the call's
    # number travels in the run headers: a replayed graph bakes its arguments in)
    c = pocs_amd.Context(0)
    c.configure(plan, env, K=K, N=N, seed=seed)
    c.set_batch(batch)
    c.set_shard(rank * n_local, n_local)
    par.connect_contexts(c, dist, rank, world)
    whole = {}
    for tag, groups in (("one", 1), ("sub2", 2), ("again", 0)):
        c.set_option(pocs_amd.OPT_SUB_BATCHES, groups)
        c.set_seed(seed)
        c.run_gmm_estimation()
        torch.cuda.synchronize()
        whole[tag] = (list(c.batch_probabilities()), np.array([c.moments(w, K) for w in range(56)]))
        dist.barrier()
    c.run_gmm_estimation()                                # (no rewind: the next runs of the context)
    torch.cuda.synchronize()
    whole["next"] = (list(c.batch_probabilities()), None)
    wait_us = c.exchange_wait_us()
    assert 0.0 <= wait_us[0] <= wait_us[1] <= wait_us[2] < 5e6
    dist.barrier()
    c.close()
    # skew tolerance: TWO engines (two batches of runs, two streams, two exchange buffers) in flight through the in-tail
    # exchange, against the same two batches through the collective
    pair = {}
    for mode in ("gloo2", "fused2"):
        cs, es = [], []
        for i in range(2):
            c = pocs_amd.Context(0)
            c.configure(plan, env, K=K, N=N, seed=seed + 10 + i)
            e = par.GpuEngine(c, 56, K, N, rank=rank, world=world, per_rank=n_local, batch=batch, stream=torch.cuda.Stream())
            if mode == "fused2":
                e.connect_onehop(dist, rank, world)
            cs.append(c); es.append(e)
        (par.run_gmm_pipelined(es, dist) if mode == "gloo2" else par.run_gmm_onehop_fused(es))
        torch.cuda.synchronize()
        pair[mode] = list(es[0].probabilities()) + list(es[1].probabilities())
        dist.barrier()
        for c in cs:
            c.close()
    res["gloo2"], res["fused2"] = pair["gloo2"], pair["fused2"]
    # an ODD number of waypoints, two calls on the same connected buffers: the last exchange of the first call and
    # the first of the second must not share a slot set (slots alternate with a running exchange count, not with
    # the waypoint's parity)
    odd = dict(traj=plan["traj"][:21], odom=plan["odom"][:20])
    for mode in ("gloo_odd", "fused_odd", "onehop_odd"):
        c = pocs_amd.Context(0)
        c.configure(odd, env, K=K, N=N, seed=seed)
        e = par.GpuEngine(c, 21, K, N, rank=rank, world=world, per_rank=n_local, batch=batch, stream=torch.cuda.Stream())
        if mode != "gloo_odd":
            e.connect_onehop(dist, rank, world)
        both = []
        for call in range(3):
            if mode == "gloo_odd":
                par.run_gmm_pipelined([e], dist)
            elif mode == "fused_odd":
                par.run_gmm_onehop_fused([e])
            else:
                par.run_gmm_onehop([e])
            torch.cuda.synchronize()
            both += list(e.probabilities())
        res[mode] = both
        dist.barrier()
        c.close()
    if rank == 0:
        np.savez(out, p_gloo=res["gloo"][0], p_one=res["onehop"][0], m_gloo=res["gloo"][1], m_one=res["onehop"][1],
                 p_again=res["onehop_again"][0], p_fused=res["fused"][0], m_fused=res["fused"][1],
                 p_whole=whole["one"][0], m_whole=whole["one"][1], p_whole2=whole["sub2"][0], m_whole2=whole["sub2"][1],
                 p_whole3=whole["again"][0], p_whole_next=whole["next"][0],
                 odd_gloo=res["gloo_odd"], odd_fused=res["fused_odd"], odd_onehop=res["onehop_odd"],
                 two_gloo=res["gloo2"], two_fused=res["fused2"])
    dist.destroy_process_group()


def test_onehop_exchange_equals_the_collective(tmp_path, pocs, plan, env):
    """pocs_gmm_exchange_local (IPC-mapped slots, rank-order sum, advance in the same launch) on two
    processes sharing ONE card: bitwise what the gloo all-reduce path gives (two ranks: a + b either
    way), run 0's moments of every waypoint included; and a second call on the same buffers redraws."""
    import torch.multiprocessing as mp
    n_local, K, seed, batch = 6000, 3, 91, 3
    out = tmp_path / "res.npz"
    port = 29900 + (os.getpid() % 90)
    mp.spawn(_worker_onehop, args=(2, port, n_local, K, seed, batch, str(out)), nprocs=2, join=True)
    got = np.load(out)
    assert list(got["p_one"]) == list(got["p_gloo"])
    assert np.array_equal(got["m_one"], got["m_gloo"])
    assert list(got["p_again"]) != list(got["p_one"]) and all(0 < p < 1 for p in got["p_again"])
    # the exchange in the sampling launch's tail: the same bits again
    assert list(got["p_fused"]) == list(got["p_gloo"]) and np.array_equal(got["m_fused"], got["m_gloo"])
    # the whole call in one library call (graph replay, exchange in the tails): the same bits -- one launch per waypoint, two
    # sub-batches, a third call after a rewind; and the call after that redraws
    assert list(got["p_whole"]) == list(got["p_gloo"]) and np.array_equal(got["m_whole"], got["m_gloo"])
    assert list(got["p_whole2"]) == list(got["p_gloo"]) and np.array_equal(got["m_whole2"], got["m_gloo"])
    assert list(got["p_whole3"]) == list(got["p_gloo"])
    assert list(got["p_whole_next"]) != list(got["p_gloo"]) and all(0 < p < 1 for p in got["p_whole_next"])
    # 21 waypoints, three calls in a row on the same buffers: every call of both one-hop forms equals the collective's
    assert list(got["odd_fused"]) == list(got["odd_gloo"]) and list(got["odd_onehop"]) == list(got["odd_gloo"])
    assert len(set(got["odd_gloo"])) == len(got["odd_gloo"])
    # two engines in flight (bench.py POCS_ENGINES=2): the same bits as the collective, batch by batch
    assert list(got["two_fused"]) == list(got["two_gloo"]) and len(got["two_gloo"]) == 2 * batch
    with pocs.Context(0) as c:                              # and both equal one process on the whole mixture
        c.configure(plan, env, K=K, N=2 * n_local, seed=seed)
        c.set_batch(batch)
        c.run_gmm_estimation()
        assert np.allclose(got["p_one"], c.batch_probabilities(), rtol=0, atol=1e-12)


def test_two_processes_one_gpu_equal_one_process(tmp_path, pocs, plan, env):
    import torch.multiprocessing as mp
    n_local, K, seed, batch = 5000, 3, 77, 2
    out = tmp_path / "res.npy"
    port = 29700 + (os.getpid() % 200)
    mp.spawn(_worker, args=(2, port, n_local, K, seed, batch, str(out)), nprocs=2, join=True)
    got = np.load(out)
    N = 2 * n_local
    want, want_mc = [], []
    with pocs.Context(0) as c:
        for s in (seed, seed + 1):
            c.configure(plan, env, K=K, N=N, seed=s)
            c.set_batch(batch)
            c.run_gmm_estimation()
            want += list(c.batch_probabilities())
        c.configure(plan, env, K=K, N=N, seed=seed)
        c.set_batch(batch)
        c.run_simulation()
        want_mc = list(c.batch_probabilities())
    assert np.allclose(got[:4], want, rtol=0, atol=1e-12)         # two ranks' moment sums added in another order
    assert list(got[4:6]) == want_mc and got[6] == want_mc[0]      # integer counts: exact


def test_cfg4_full_size_eight_shards_on_one_card(pocs, plan, env):
    """BASELINE.json configs[3] at full size -- 10^7 samples, 8 components, 500 waypoints, sharded over EIGHT ranks
    -- rehearsed on the one card of this box: eight contexts in one process, each with the shard `shard_range` gives
    its rank (1.25 * 10^6 samples), walk the protocol of parallel.run_gmm_sharded in lockstep; the all-reduce between
    two waypoints is done here (the eight moment buffers added in rank order, the sum written back to all eight).
    Against configs[2] -- the same workload on one GPU in one context: the shards' sums are added in another order
    than the single tree, so the bits may differ in the last place; the survivor counts of every waypoint and
    component must be equal and the probabilities within the north star's 1e-6.  What this does
    NOT exercise is RCCL and xGMI between eight physical GPUs."""
    import torch
    from importlib import import_module
    par = import_module("probability-of-collision-for-safe-planning_amd.parallel")
    big = pocs.resample_plan(plan, 500)
    N, K, W, world, seed = 10_000_000, 8, 500, 8, 4242
    ctxs, engs = [], []
    side = torch.cuda.Stream()                                    # the eight contexts' launches and the sums, in order on one stream
    with torch.cuda.stream(side):
        for r in range(world):
            c = pocs.Context(0)
            c.configure(big, env, K=K, N=N, seed=seed)
            c.set_option(pocs.OPT_STORE_SAMPLES, 0)
            ctxs.append(c)
            engs.append(par.GpuEngine(c, W, K, N, rank=r, world=world, stream=side))
        assert sum(e.count for e in engs) == N and engs[0].count == 1_250_000
        for e in engs:
            e.begin()
        for w in range(W):
            for e in engs:
                e.step_local(w)
            acc = engs[0].moments(w).clone()
            for e in engs[1:]:
                acc += e.moments(w)                               # rank order, like the library's own exchange
            for e in engs:
                e.moments(w).copy_(acc)
        sharded = [e.end() for e in engs]
    torch.cuda.synchronize()
    assert len(set(sharded)) == 1                                 # every rank ends with the same probability
    m_sh = np.array([ctxs[3].moments(w, K) for w in range(W)])
    p_sh = ctxs[3].waypoint_probabilities().copy()
    for c in ctxs:
        c.close()
    with pocs.Context(0) as c:
        c.configure(big, env, K=K, N=N, seed=seed)
        c.set_option(pocs.OPT_STORE_SAMPLES, 0)
        one = c.run_gmm_estimation()
        m_one = np.array([c.moments(w, K) for w in range(W)])
        p_one = c.waypoint_probabilities().copy()
    assert np.array_equal(m_sh[..., :2], m_one[..., :2])          # survivors and collisions, every waypoint, every component
    assert np.max(np.abs(p_sh - p_one)) <= 1e-6 and abs(sharded[0] - one) <= 1e-6
    # the sums: differences of the last place of the LARGEST term (a sum like S_xy nearly cancels), i.e. relative to
    # the column's scale sum |x y| <= sqrt(S_xx S_yy)
    scale = np.maximum(np.abs(m_one), np.sqrt(np.abs(m_one[..., [0, 1, 5, 8, 10, 5, 5, 5, 8, 8, 10]] * m_one[..., [0, 1, 0, 0, 0, 5, 8, 10, 8, 10, 10]])))
    worst = np.max(np.abs(m_sh - m_one) / np.maximum(scale, 1e-300))
    print("cfg4 rehearsal: sharded %.17g, one GPU %.17g, |dp| max %.3g, largest moment difference relative to its column's scale %.3g"
          % (sharded[0], one, np.max(np.abs(p_sh - p_one)), worst))
    assert worst < 1e-6                                           # (observed 8e-10: last-place differences fed back through 500 truncations)
    assert 0.0 < one < 1.0 and m_one[:, :, 1].sum() > 0


def test_engines_on_the_default_stream_are_ordered_with_torch(pocs, plan, env):
    """A GpuEngine built without a stream runs on torch's current stream -- by default the null stream, whose handle
    is 0, which pocs_set_stream reads as "the context's own non-blocking stream": the engine's launches and the
    caller's collectives on the current stream were then not ordered at all (every rank of the rehearsal above ended
    with a different probability).  GpuEngine now names the null stream (hipStreamLegacy).  Two shards in one
    process, the all-reduce done with torch operations on the current stream: the same bits as with an explicit
    stream, and the same on both ranks."""
    import torch
    from importlib import import_module
    par = import_module("probability-of-collision-for-safe-planning_amd.parallel")
    N, K, W, world = 400_000, 3, 56, 2
    res = {}
    for mode in ("default", "explicit"):
        side = torch.cuda.Stream() if mode == "explicit" else None
        ctxs, engs = [], []
        with (torch.cuda.stream(side) if side is not None else torch.cuda.stream(torch.cuda.current_stream())):
            for r in range(world):
                c = pocs.Context(0)
                c.configure(plan, env, K=K, N=N, seed=99)
                ctxs.append(c)
                engs.append(par.GpuEngine(c, W, K, N, rank=r, world=world, stream=side))
            for e in engs:
                e.begin()
            for w in range(W):
                for e in engs:
                    e.step_local(w)
                acc = engs[0].moments(w) + engs[1].moments(w)
                for e in engs:
                    e.moments(w).copy_(acc)
            ps = [e.end() for e in engs]
        torch.cuda.synchronize()
        res[mode] = (ps, np.array([ctxs[1].moments(w, K) for w in range(W)]))
        for c in ctxs:
            c.close()
    assert res["default"][0][0] == res["default"][0][1] == res["explicit"][0][0] == res["explicit"][0][1]
    assert np.array_equal(res["default"][1], res["explicit"][1])
    assert 0.0 < res["default"][0][0] < 1.0
