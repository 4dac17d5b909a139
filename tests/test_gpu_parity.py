"""Parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Everything is compared BIT FOR BIT: integer work (collision flags, hit counters,
survivor counts), samples and particles, and -- since numerics v7 fixes the summation tree of the
moment sums, which the oracle restates -- the floating-point sums, hence every mixture state and every
probability of a free-running estimation, at any sample count and for any number of runs per launch.
(The north star's tolerance for the GMM path is 1e-6 absolute; `==` is stricter.)"""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x5EED0001
WEYL = 0x9E3779B97F4A7C15            # effective seed of the r-th run of a context = seed + r * WEYL (mod 2^64)


@pytest.fixture(scope="module")
def ctx(pocs):
    c = pocs.Context(0)
    yield c
    c.close()


def close_moments(got, want):
    assert np.array_equal(got[..., :2], want[..., :2]), "survivor / collision counts differ"
    assert np.array_equal(got, want), "moment sums differ"


def test_host_chain_is_bitwise_the_oracles(ctx, orc, plan, env):
    cfg = orc.config(plan, env, K=3)
    ctx.configure(plan, env, K=3, N=64, seed=SEED)
    ctx.run_gmm_estimation()
    got, want = ctx.host_chain(8), orc.host_chain(cfg, SEED)
    for k in ("applied", "noisy", "z", "mu", "cov"):
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize("N", [1, 63, 257, 4096])
def test_mc_hits_bit_exact(ctx, orc, plan, env, N):
    cfg = orc.config(plan, env, K=3)
    ctx.configure(plan, env, K=3, N=N, seed=SEED)
    p = ctx.run_simulation()
    n, hits, parts = orc.run_mc(cfg, SEED, N, want_particles=True)
    xyz, got_hits = ctx.particles(N)
    assert np.array_equal(got_hits, hits)                     # particlecollisions, u32 per particle
    assert np.array_equal(xyz, parts)                         # final mcparticles, bit for bit
    assert p == n / N


def test_cfg5_mc_rollouts_full_size(ctx, pocs, orc, plan, env):
    """BASELINE.json configs[4] at full size: 10^5 roll-outs x 500 waypoints (per-particle footprint
    check at every waypoint).  Hit counters and final particles bit for bit against the oracle
    (~5 s of CPU), for the streaming kernels and the in-register variant."""
    big = pocs.resample_plan(plan, 500)
    N = 100_000
    cfg = orc.config(big, env, K=1)
    n, hits, parts = orc.run_mc(cfg, SEED, N, want_particles=True)
    for fused in (0, 1):
        ctx.configure(big, env, K=1, N=N, seed=SEED)
        ctx.set_option(pocs.OPT_MC_FUSED, fused)
        p = ctx.run_simulation()
        xyz, got_hits = ctx.particles(N)
        ctx.set_option(pocs.OPT_MC_FUSED, 0)
        assert p == n / N
        assert np.array_equal(got_hits, hits) and np.array_equal(xyz, parts)
    assert hits.max() > 50 and 0 < n < N                       # a roll-out that stays in collision for many waypoints


def test_cfg1_exact_shape(ctx, orc, plan, env):
    """BASELINE.json configs[0] exactly: bundled plan, 1k particles, K = 3 -- every stage against the
    oracle (this is the shape at which components get close to running out of survivors)."""
    cfg = orc.config(plan, env, K=3)
    for seed in (SEED, SEED + 1, SEED + 2):
        ctx.configure(plan, env, K=3, N=1000, seed=seed)
        p = ctx.run_gmm_estimation()
        want = orc.run_gmm(cfg, seed, 1000, want_samples=True)
        got_m = np.array([ctx.moments(w, 3) for w in range(56)])
        assert np.array_equal(got_m, want["moments"])
        assert np.array_equal(ctx.waypoint_probabilities(), want["probs"]) and p == want["prob"]
        assert np.array_equal(ctx.gmm_samples(1000)[1], want["flags"])
        ctx.set_seed(seed)
        assert ctx.run_simulation() == orc.run_mc(cfg, seed, 1000)[0] / 1000


def test_mc_fused_equals_streaming(ctx, pocs, plan, env):
    ctx.configure(plan, env, K=3, N=5000, seed=7)
    p1 = ctx.run_simulation()
    x1, h1 = ctx.particles(5000)
    ctx.set_option(pocs.OPT_MC_FUSED, 1)
    ctx.set_seed(7)
    p2 = ctx.run_simulation()
    x2, h2 = ctx.particles(5000)
    ctx.set_option(pocs.OPT_MC_FUSED, 0)
    assert p1 == p2 and np.array_equal(h1, h2) and np.array_equal(x1, x2)


def test_gmm_first_waypoint_samples_bit_exact(ctx, orc, plan, env):
    """W = 1: the mixture is the initial one, so samples, flags and counts must match bit for bit."""
    one = dict(traj=plan["traj"][:1], odom=plan["odom"][:0])
    cfg = orc.config(one, env, K=3)
    N = 5000
    ctx.configure(one, env, K=3, N=N, seed=SEED)
    p = ctx.run_gmm_estimation()
    want = orc.run_gmm(cfg, SEED, N, want_samples=True)
    xyz, flags = ctx.gmm_samples(N)
    assert np.array_equal(flags, want["flags"])
    assert np.array_equal(xyz, want["samples"])
    close_moments(ctx.moments(0, 3), want["moments"][0])
    assert p == want["prob"]


def test_gmm_two_million_samples_bit_exact(ctx, orc, plan, env):
    """Same at N = 2*10^6 (3*10^6 Box-Muller radii through the device's own sqrt sequence,
    pocs_sqrt_radius2, against the host's correctly rounded sqrt)."""
    one = dict(traj=plan["traj"][:1], odom=plan["odom"][:0])
    cfg = orc.config(one, env, K=2)
    N = 2_000_000
    ctx.configure(one, env, K=2, N=N, seed=SEED + 99)
    ctx.run_gmm_estimation()
    want = orc.run_gmm(cfg, SEED + 99, N, want_samples=True)
    xyz, flags = ctx.gmm_samples(N)
    assert np.array_equal(flags, want["flags"])
    assert np.array_equal(xyz, want["samples"])
    assert np.array_equal(ctx.moments(0, 2), want["moments"][0])


@pytest.mark.parametrize("K,N", [(1, 3000), (3, 10000), (8, 6000), (3, 100)])
def test_gmm_matches_oracle(ctx, orc, plan, env, K, N):
    cfg = orc.config(plan, env, K=K)
    seed = SEED + K
    ctx.configure(plan, env, K=K, N=N, seed=seed)
    p = ctx.run_gmm_estimation()
    W = cfg.W
    got_m = np.array([ctx.moments(w, K) for w in range(W)])
    got_s = np.array([ctx.gmm_state_raw(w, K) for w in range(W)])
    chain = orc.host_chain(cfg, seed)

    # (1) stage by stage, the oracle fed with the GPU's own inputs: every stage must agree to the last bit
    st0 = orc.gmm_advance(cfg, orc.gmm_initial_state(cfg), None)
    assert np.array_equal(got_s[0][:, :14], st0[:, :14])
    for w in range(W):
        mom, samples, flags, comp = orc.gmm_waypoint(cfg, seed, w, got_s[w], 0, N, want_samples=True)
        assert np.array_equal(got_m[w], mom), w                            # survivors / collisions / the nine sums
        if w + 1 < W:                                                      # device EKF / truncation / Cholesky
            nxt = orc.gmm_advance(cfg, got_s[w], got_m[w], chain["applied"][w], chain["Mdiag"][w], chain["z"][w])
            assert np.array_equal(got_s[w + 1][:, :14], nxt[:, :14]), w
    xyz, gflags = ctx.gmm_samples(N)                                       # last waypoint, as stored in HBM
    assert np.array_equal(gflags, flags) and np.array_equal(xyz, samples)

    # (2) free running: both sides from the seed alone, and still every bit
    want = orc.run_gmm(cfg, seed, N)
    assert np.array_equal(got_m, want["moments"])
    assert np.array_equal(ctx.waypoint_probabilities(), want["probs"])
    assert p == want["prob"]
    assert np.array_equal(got_s[..., :14], want["states"][..., :14])


def test_gmm_one_million_samples_bit_exact(ctx, orc, plan, env):
    """configs[1] of BASELINE.json at full size: bundled plan, 3 components, 10^6 samples, free running:
    within the north star's 1e-6 -- in fact equal."""
    cfg = orc.config(plan, env, K=3)
    N = 1000000
    ctx.configure(plan, env, K=3, N=N, seed=SEED)
    p = ctx.run_gmm_estimation()
    want = orc.run_gmm(cfg, SEED, N)
    assert abs(p - want["prob"]) <= 1e-6                       # the stated tolerance
    assert p == want["prob"] and np.array_equal(ctx.waypoint_probabilities(), want["probs"])
    got_m = np.array([ctx.moments(w, 3) for w in range(cfg.W)])
    assert np.array_equal(got_m, want["moments"])


def _run_of_batch(c, r, K, W):
    c.select_batch_run(r)
    return (np.array([c.moments(w, K) for w in range(W)]), c.waypoint_probabilities().copy(),
            np.array([c.gmm_state_raw(w, K) for w in range(W)])[..., :14])


@pytest.mark.parametrize("R", [20, 64])
def test_timed_launch_shapes_against_the_oracle(pocs, orc, plan, env, R):
    """The shapes bench.py times -- 10^6 samples, K = 3, 20 runs per call (the driver's invocation) and 64
    (the default) -- compared with the oracle: run 0 and run R - 1 of the batch on their effective seeds,
    every waypoint's moments, probabilities and mixture, bit for bit; and the whole batch through
    run-ahead (one command per run, the OpenRAVE adapter's default) gives the same bits again."""
    N, K = 1_000_000, 3
    cfg = orc.config(plan, env, K=K)
    with pocs.Context(0) as c:
        c.configure(plan, env, K=K, N=N, seed=SEED)
        c.set_batch(R)
        p0 = c.run_gmm_estimation()
        finals = list(c.batch_probabilities())
        got = {r: _run_of_batch(c, r, K, cfg.W) for r in (0, 1, R - 1)}
        for r in (0, R - 1):
            want = orc.run_gmm(cfg, (SEED + r * WEYL) % 2**64, N)
            assert np.array_equal(got[r][0], want["moments"]), r
            assert np.array_equal(got[r][1], want["probs"]) and finals[r] == want["prob"], r
            assert np.array_equal(got[r][2], want["states"][..., :14]), r
        assert p0 == finals[0]
        c.set_batch(1)
        c.set_option(pocs.OPT_RUN_AHEAD, 64)
        c.set_seed(SEED)
        for r in range(2):
            assert c.run_gmm_estimation() == finals[r]
            assert np.array_equal(np.array([c.moments(w, K) for w in range(cfg.W)]), got[r][0]), r
            assert np.array_equal(c.waypoint_probabilities(), got[r][1]), r


@pytest.mark.parametrize("K,N,R", [(3, 300000, 7), (8, 150000, 3), (1, 2_000_000, 1), (2, 40001, 20), (3, 5001, 33)])
def test_runs_per_launch_do_not_change_a_bit(pocs, plan, env, K, N, R):
    """The moment sums are defined on a run's virtual slices, not on the launch: R runs per call (blocks
    that work on several slices, some of them on two runs), one run per call and eager launches give
    the same moments, states and samples for every run, and the same again when repeated."""
    with pocs.Context(0) as c:
        c.configure(plan, env, K=K, N=N, seed=123)
        singles = []
        for r in range(min(R, 3)):
            c.run_gmm_estimation()
            singles.append(_run_of_batch(c, 0, K, 56) + (c.gmm_samples(N),))
        c.set_batch(R)
        for rep in range(2):
            if rep == 1:
                c.set_option(pocs.OPT_USE_GRAPH, 0)
            c.set_seed(123)
            c.run_gmm_estimation()
            for r in range(min(R, 3)):
                got = _run_of_batch(c, r, K, 56)
                assert all(np.array_equal(a, b) for a, b in zip(got, singles[r][:3])), (rep, r)
                x, f = c.gmm_samples(N)
                assert np.array_equal(f, singles[r][3][1]) and np.array_equal(x, singles[r][3][0]), (rep, r)


@pytest.mark.parametrize("K,N", [(3, 300000), (8, 150000), (1, 1_000_001), (2, 3001)])
def test_lone_call_changes_no_bit(pocs, orc, plan, env, K, N):
    """One run per call has a launch form of its own (POCS_OPT_LONE_CALL, the default: every block of waypoint w's
    launch adds the rows of w - 1 and advances the mixture in its head; no tickets, no closing block).  Against the
    ticket form, replayed from a graph and launched eagerly, twice over: moments, probabilities, mixture states of
    all 56 waypoints, the last waypoint's samples and flags -- and the oracle's probabilities."""
    with pocs.Context(0) as c:
        c.configure(plan, env, K=K, N=N, seed=77)
        outs = []
        for lone, graph in ((1, 1), (0, 1), (1, 0), (0, 0), (1, 1)):
            c.set_option(pocs.OPT_LONE_CALL, lone)
            c.set_option(pocs.OPT_USE_GRAPH, graph)
            c.set_seed(77)
            p = c.run_gmm_estimation()
            outs.append((np.float64(p),) + _run_of_batch(c, 0, K, 56) + tuple(c.gmm_samples(N)))
        for i, o in enumerate(outs[1:]):
            assert all(np.array_equal(a, b) for a, b in zip(o, outs[0])), i + 1
    if N <= 300000:
        want = orc.run_gmm(orc.config(plan, env, K=K), 77, N)
        assert outs[0][0] == want["prob"] and np.array_equal(outs[0][2], want["probs"])


def test_gmm_variants_agree(ctx, pocs, plan, env):
    """Graph replay vs eager launches, with and without the sample store, with the profiling events,
    and the per-waypoint step API: the same units with the same arithmetic, so everything is bitwise
    the same."""
    ctx.configure(plan, env, K=3, N=20000, seed=11)
    base = ctx.run_gmm_estimation()
    base_probs = ctx.waypoint_probabilities().copy()
    base_m = ctx.moments(30, 3).copy()
    for opt, val in ((pocs.OPT_USE_GRAPH, 0), (pocs.OPT_STORE_SAMPLES, 0), (pocs.OPT_PROFILE, 1)):
        ctx.set_option(opt, val)
        ctx.set_seed(11)
        assert ctx.run_gmm_estimation() == base
        assert np.array_equal(ctx.waypoint_probabilities(), base_probs)
        assert np.array_equal(ctx.moments(30, 3), base_m)        # fixed-shape reduction: bitwise
        if opt == pocs.OPT_PROFILE:
            ms, n = ctx.kernel_time()
            assert n == 56 and ms > 0                            # one launch per waypoint
        ctx.set_option(opt, 1 - val)
    ctx.set_seed(11)
    again = ctx.run_gmm_estimation()                             # graph replayed a second time
    assert again == base
    ctx.set_seed(11)
    ctx.gmm_begin()
    for w in range(56):
        ctx.gmm_step_local(w)
    assert ctx.gmm_end() == base
    with pytest.raises(pocs.PocsError):                          # the queue-driven kernel of round 2 is retired
        ctx.set_option(pocs.OPT_PERSISTENT, 1)
    ctx.set_option(pocs.OPT_PERSISTENT, 0)


@pytest.mark.parametrize("mc", [False, True])
def test_graph_replays_survive_readbacks(pocs, plan, env, mc):
    """A caller that reads mixture states and samples back between two runs (dozens of small synchronous
    copies plus a large one, results kept) and then runs again: every replay of the captured graph must give
    what a fresh context gives.  (With memset / memcpy nodes inside the graph, ROCm 7.2 lost the third
    replay of exactly this sequence -- round 2's library included; the graphs hold kernel nodes only now.)"""
    K, N = 3, 300000
    def sequence(graph):
        out, keep = [], []
        with pocs.Context(0) as c:
            c.configure(plan, env, K=K, N=N, seed=123)
            c.set_num_particles(N)
            c.set_option(pocs.OPT_USE_GRAPH, graph)
            for i in range(6):
                out.append(c.run_simulation() if mc else c.run_gmm_estimation())
                if mc:
                    keep.append(c.particles(N))
                else:
                    keep.append((np.array([c.gmm_state_raw(w, K) for w in range(56)]), c.gmm_samples(N)))
        return out
    assert sequence(1) == sequence(0)


def test_sub_batches_on_side_streams_change_no_bit(tmp_path):
    """POCS_OPT_SUB_BATCHES = 2: a call's runs issued as two sub-batches on streams of their own (graph forked and
    joined by events; eager too), their launches overlapping: every run's probabilities and moments as with one
    launch per waypoint for all of them.  Three and more lost on the clock and are refused."""
    import hashlib
    import pocs_amd
    plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
    digests = {}
    for g in (1, 2):
        h = hashlib.sha256()
        with pocs_amd.Context(0) as c:
            for graph in (1, 0):
                c.configure(plan, env, K=3, N=100001, seed=5)
                c.set_option(pocs_amd.OPT_USE_GRAPH, graph)
                c.set_option(pocs_amd.OPT_SUB_BATCHES, g)
                c.set_batch(17)
                for rep in range(2):
                    c.run_gmm_estimation()
                    h.update(np.array(c.batch_probabilities()).tobytes())
                    for r in (0, 8, 16):
                        c.select_batch_run(r)
                        h.update(np.array([c.moments(w, 3) for w in range(56)]).tobytes())
            with pytest.raises(pocs_amd.PocsError):
                c.set_option(pocs_amd.OPT_SUB_BATCHES, 3)
        digests[g] = h.hexdigest()
    assert digests[1] == digests[2], digests


def test_batch_equals_consecutive_single_runs(ctx, plan, env):
    """R estimations advanced in lockstep (one launch per waypoint for all of them) must give,
    bit for bit, what R consecutive single runs give."""
    ctx.configure(plan, env, K=3, N=5001, seed=17)
    seq, seq_probs = [], []
    for _ in range(5):
        seq.append(ctx.run_gmm_estimation())
        seq_probs.append(ctx.waypoint_probabilities().copy())
    m0 = None
    ctx.set_seed(17)
    ctx.run_gmm_estimation()
    m0 = ctx.moments(40, 3).copy()
    ctx.set_seed(17)
    ctx.set_batch(5)
    try:
        p0 = ctx.run_gmm_estimation()
        assert p0 == seq[0] and list(ctx.batch_probabilities()) == seq
        assert np.array_equal(ctx.waypoint_probabilities(), seq_probs[0])
        assert np.array_equal(ctx.moments(40, 3), m0)
        xyz, flags = ctx.gmm_samples(5001)                      # run 0's last waypoint
        assert len(flags) == 5001
        ctx.set_seed(17)
        ctx.gmm_begin()                                          # the sharded protocol, batched
        assert ctx.gmm_moments_len() == 5 * 33
        for w in range(56):
            ctx.gmm_step_local(w)
        assert ctx.gmm_end() == seq[0] and list(ctx.batch_probabilities()) == seq
        assert ctx.run_gmm_estimation() != seq[0]                # the next batch redraws
    finally:
        ctx.set_batch(1)
    ctx.set_seed(17)
    assert ctx.run_gmm_estimation() == seq[0]


def test_mc_batch_equals_consecutive_single_runs(ctx, pocs, plan, env):
    N = 3001
    ctx.configure(plan, env, K=3, N=N, seed=23)
    seq = [ctx.run_simulation() for _ in range(4)]
    ctx.set_seed(23)
    ctx.run_simulation()
    x0, h0 = ctx.particles(N)
    for fused in (0, 1):
        ctx.set_option(pocs.OPT_MC_FUSED, fused)
        ctx.set_seed(23)
        ctx.set_batch(4)
        try:
            assert ctx.run_simulation() == seq[0]
            assert list(ctx.batch_probabilities()) == seq
            assert ctx.mc_batch_counts() == [round(p * N) for p in seq]
            x, h = ctx.particles(N)                               # run 0 of the batch
            assert np.array_equal(x, x0) and np.array_equal(h, h0)
        finally:
            ctx.set_batch(1)
    ctx.set_option(pocs.OPT_MC_FUSED, 0)


def test_pipelined_engines_match_single_runs(pocs, plan, env):
    """parallel.GpuEngine / run_gmm_pipelined (what bench.py runs per rank over RCCL), here without
    a process group: two engines on two torch streams, batches of 2, must reproduce the four
    corresponding single runs bit for bit."""
    import torch
    from importlib import import_module
    par = import_module("probability-of-collision-for-safe-planning_amd.parallel")
    N, K, W = 6000, 3, 56
    want = []
    with pocs.Context(0) as c:
        for seed in (5, 6):
            c.configure(plan, env, K=K, N=N, seed=seed)
            want.append([c.run_gmm_estimation(), c.run_gmm_estimation()])
    ctxs, engs = [], []
    try:
        for seed in (5, 6):
            c = pocs.Context(0)
            c.configure(plan, env, K=K, N=N, seed=seed)
            ctxs.append(c)
            engs.append(par.GpuEngine(c, W, K, N, rank=0, world=1, per_rank=N, batch=2, stream=torch.cuda.Stream()))
        first = par.run_gmm_pipelined(engs, None)
        torch.cuda.synchronize()
        assert first == [want[0][0], want[1][0]]
        assert [list(e.probabilities()) for e in engs] == want
        assert engs[0].moments(3).shape == (2 * K * 11,)
        # a smaller last call on an engine sized for more
        engs[0].set_batch(1)
        ctxs[0].set_seed(5)
        assert par.run_gmm_pipelined(engs[:1], None) == [want[0][0]]
    finally:
        for c in ctxs:
            c.close()


def test_runs_redraw_and_seed_rewinds(ctx, plan, env):
    ctx.configure(plan, env, K=3, N=4000, seed=5)
    a, b = ctx.run_gmm_estimation(), ctx.run_gmm_estimation()
    assert a != b                                # a fresh stream per run, as the reference redraws
    ctx.set_seed(5)
    assert ctx.run_gmm_estimation() == a and ctx.run_gmm_estimation() == b


def test_shards_partition_the_work(ctx, orc, plan, env):
    """Two shards of one mixture waypoint add up to the whole (what the all-reduce relies on)."""
    cfg = orc.config(plan, env, K=3)
    N = 9001
    ctx.configure(plan, env, K=3, N=N, seed=21)
    ctx.run_gmm_estimation()
    whole = ctx.moments(0, 3)
    parts = []
    for first, count in ((0, 4000), (4000, 5001)):
        ctx.set_shard(first, count)
        ctx.set_seed(21)
        ctx.gmm_begin()
        for w in range(56):
            ctx.gmm_step_local(w)
        ctx.gmm_end()
        # without the exchange only waypoint 0 is meaningful: its moments are this shard's sums
        parts.append(ctx.moments(0, 3).copy())
        want = orc.gmm_waypoint(cfg, 21, 0, orc.gmm_advance(cfg, orc.gmm_initial_state(cfg), None), first, count, n_total=N)
        close_moments(parts[-1], want)
    ctx.set_shard()
    assert np.array_equal((parts[0] + parts[1])[:, :2], whole[:, :2])
    assert np.allclose(parts[0] + parts[1], whole, rtol=1e-12)
    ctx.set_shard(0, 1000)
    ctx.set_seed(21)
    n0 = ctx.mc_run_local()
    ctx.set_shard(1000, N - 1000)
    ctx.set_seed(21)
    n1 = ctx.mc_run_local()
    ctx.set_shard()
    ctx.set_seed(21)
    assert n0 + n1 == ctx.mc_run_local() == orc.run_mc(cfg, 21, N)[0]


@pytest.mark.parametrize("first,count", [(0, 20001), (2 * (1024 + 512 + 100), 7001), (2 * 1023, 2), (2 * 511, 4100),
                                         (4096 + 2 * 37, 1), (10000, 10001)])
def test_shards_anywhere(ctx, orc, plan, env, first, count):
    """A shard may start at any even index and have any length: its samples, flags and moments of
    waypoint 0 are the oracle's for exactly that index range (draws are keyed by the global index)."""
    K, N = 3, 20001
    cfg = orc.config(plan, env, K=K)
    ctx.configure(plan, env, K=K, N=N, seed=19)
    ctx.set_shard(first, count)
    try:
        ctx.gmm_begin()
        ctx.gmm_step_local(0)
        xyz, flags = ctx.gmm_samples(count)
        for w in range(1, 56):
            ctx.gmm_step_local(w)
        ctx.gmm_end()
        got = ctx.moments(0, K)
    finally:
        ctx.set_shard()
    st0 = orc.gmm_advance(cfg, orc.gmm_initial_state(cfg), None)
    mom, samples, oflags, _ = orc.gmm_waypoint(cfg, 19, 0, st0, first, count, want_samples=True, n_total=N)
    assert np.array_equal(flags, oflags) and np.array_equal(xyz, samples)
    close_moments(got, mom)


@pytest.mark.parametrize("K,N", [(3, 2), (3, 3), (8, 9), (2, 1), (8, 40)])
def test_degenerate_mixtures_match_oracle(ctx, orc, plan, env, K, N):
    """Tiny N: components run out of survivors (< 2) or all samples collide -- cases the reference
    leaves undefined; the build retires such components, on both sides alike."""
    cfg = orc.config(plan, env, K=K)
    for seed in (1, 2, 3):
        ctx.configure(plan, env, K=K, N=N, seed=seed)
        p = ctx.run_gmm_estimation()
        want = orc.run_gmm(cfg, seed, N)
        assert np.array_equal(ctx.waypoint_probabilities(), want["probs"])
        assert p == want["prob"]
        got_alive = np.array([ctx.gmm_state(w, K)[3] for w in range(cfg.W)])
        assert np.array_equal(got_alive, want["states"][:, :, 13])
        assert np.array_equal(np.array([ctx.gmm_state(w, K)[2] for w in range(cfg.W)]), want["states"][:, :, 12])
    assert not np.all(want["states"][:, :, 13] == 1.0)          # something did get retired


def test_limits_of_the_boundary(ctx, pocs, orc, plan):
    """Maximum table sizes: 64 obstacles, 32 landmarks; shortest plans W = 1, 2; an empty shard."""
    rng = np.random.default_rng(8)
    boxes = np.column_stack([rng.uniform(-3.5, 3.5, 64), rng.uniform(-1.7, 1.7, 64), rng.uniform(0.02, 0.15, 64),
                             rng.uniform(0.02, 0.15, 64), rng.uniform(-3.2, 3.2, 64)])
    env64 = dict(footprint=[0.02, 0.01, 0.3, 0.2], boxes=boxes)
    lm = np.vstack([rng.uniform(-4, 4, 32), rng.uniform(-2, 2, 32)])
    params = dict(pocs.DEFAULTS, landmarks=lm.tolist())
    for W in (1, 2, 56):
        pl = dict(traj=plan["traj"][:W], odom=plan["odom"][:max(W - 1, 0)])
        cfg = orc.config(pl, env64, K=3, landmarks=lm)
        ctx.configure(pl, env64, params=params, K=3, N=3000, seed=4)
        p = ctx.run_gmm_estimation()
        want = orc.run_gmm(cfg, 4, 3000, want_samples=True)
        assert np.array_equal(ctx.gmm_samples(3000)[1], want["flags"]) and p == want["prob"]
        ctx.set_seed(4)
        assert ctx.run_simulation() == orc.run_mc(cfg, 4, 3000)[0] / 3000
    with pytest.raises(pocs.PocsError):
        ctx.set_env(dict(footprint=[0, 0, 0.3, 0.3], boxes=np.zeros((65, 5)) + 0.1))
    with pytest.raises(pocs.PocsError):
        ctx.set_landmarks(np.zeros((2, 33)))
    # an empty shard contributes nothing and does not fault
    ctx.configure(plan, env64, K=3, N=1000, seed=4)
    ctx.set_shard(1000, 0)
    assert ctx.mc_run_local() == 0
    ctx.gmm_begin()
    ctx.gmm_step_local(0)
    ctx.set_shard()


def test_randomised_configurations_match_oracle(ctx, pocs, orc, plan):
    """Twenty random worlds (rotated boxes, off-centre footprints, noise levels, landmark sets,
    sub-plans, K, N, seeds): GMM flags/counts/probabilities and MC hit counters equal the oracle's."""
    rng = np.random.default_rng(2024)
    for case in range(20):
        M = int(rng.integers(0, 12))
        boxes = np.column_stack([rng.uniform(-3.8, 3.8, M), rng.uniform(-1.8, 1.8, M), rng.uniform(0.03, 0.6, M),
                                 rng.uniform(0.03, 0.6, M), rng.choice([0.0, 0.0, 1.5707963267948966, 1.0, -0.6, 2.2], M)])
        fp = [0.0, 0.0, float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.1, 0.4))]
        if case % 3 == 0:
            fp[0], fp[1] = float(rng.uniform(-0.1, 0.1)), float(rng.uniform(-0.1, 0.1))
        env_r = dict(footprint=fp, boxes=boxes.reshape(-1, 5))
        L = int(rng.integers(1, 12))
        lm = np.vstack([rng.uniform(-4, 4, L), rng.uniform(-2, 2, L)])
        W = int(rng.integers(2, 57))
        start = int(rng.integers(0, 57 - W))
        pl = dict(traj=plan["traj"][start:start + W], odom=plan["odom"][start:start + W - 1])
        K = int(rng.integers(1, 9))
        N = int(rng.integers(50, 4000))
        seed = int(rng.integers(0, 2 ** 62))
        params = dict(pocs.DEFAULTS, landmarks=lm.tolist(), Q=float(rng.uniform(0.01, 0.1)),
                      alphas=[float(a * rng.uniform(0.3, 3)) for a in pocs.DEFAULTS["alphas"]],
                      cov0=(np.eye(3) * rng.uniform(2e-4, 4e-3)).tolist())
        cfg = orc.config(pl, env_r, K=K, alphas=params["alphas"], Q=params["Q"], landmarks=lm, cov0=params["cov0"])
        ctx.configure(pl, env_r, params=params, K=K, N=N, seed=seed)
        p = ctx.run_gmm_estimation()
        want = orc.run_gmm(cfg, seed, N, want_samples=True)
        assert np.array_equal(ctx.waypoint_probabilities(), want["probs"]), case
        assert p == want["prob"], case
        assert np.array_equal(ctx.gmm_samples(N)[1], want["flags"]), case
        ctx.set_seed(seed)
        p_mc = ctx.run_simulation()
        n_mc, hits, _ = orc.run_mc(cfg, seed, N)
        assert p_mc == n_mc / N and np.array_equal(ctx.particles(N)[1], hits), case


def test_text_channel_is_a_drop_in(ctx, pocs, orc, plan, env):
    """The exact command sequence of MCSimulation.py:154-207,238-245 over the text channel."""
    def l2s(v):
        return "".join(str(float(x)) + " " for x in v)       # list2String, MCSimulation.py:81-85
    c = pocs.Context(0)
    try:
        c.set_env(env)
        P = pocs.DEFAULTS
        assert c.SendCommand("MyCommand [0.1,4.5,7.5,4.7]") == "output"
        assert "runGMMEstimation" in c.SendCommand("help")
        c.SendCommand("setAlphas " + l2s(P["alphas"]))
        c.SendCommand("setQ " + str(P["Q"]))
        c.SendCommand("setNumLandmarks 8")
        c.SendCommand("setLandmarks " + l2s(P["landmarks"][0]) + l2s(P["landmarks"][1]))
        c.SendCommand("setNumParticles 3000")
        c.SendCommand("setInitialCovariance " + "".join(l2s(r) for r in P["cov0"]))
        c.SendCommand("setPathLength 56")
        t, o = plan["traj"].T, plan["odom"].T
        c.SendCommand("setTrajectory " + "".join(repr(float(x)) + " " for r in t for x in r))
        c.SendCommand("setOdometry " + "".join(repr(float(x)) + " " for r in o for x in r))
        c.SendCommand("setNumGaussians 3")
        c.SendCommand("setNumGMMSamples 3000")
        c.SendCommand("setSeed 99")
        p_gmm = float(c.SendCommand("runGMMEstimation"))
        c.SendCommand("setSeed 99")
        p_mc = float(c.SendCommand("runSimulation"))
    finally:
        c.close()
    cfg = orc.config(plan, env, K=3)
    assert p_gmm == orc.run_gmm(cfg, 99, 3000)["prob"]
    assert p_mc == orc.run_mc(cfg, 99, 3000)[0] / 3000


def test_error_behaviour(pocs, plan, env):
    c = pocs.Context(0)
    try:
        with pytest.raises(pocs.PocsError) as e:
            c.SendCommand("setLandmarks 1 2 3 4")
        assert e.value.code == -2                                  # before setNumLandmarks
        with pytest.raises(pocs.PocsError) as e:
            c.SendCommand("setTrajectory 1 2 3")
        assert e.value.code == -2                                  # before setPathLength
        c.SendCommand("setPathLength 2")
        with pytest.raises(pocs.PocsError) as e:
            c.SendCommand("setTrajectory 1 2 3")
        assert e.value.code == -1                                  # wrong token count
        with pytest.raises(pocs.PocsError) as e:
            c.SendCommand("setAlphas 1 2 3 4 5")
        assert e.value.code == -1                                  # the reference overflows here
        with pytest.raises(pocs.PocsError) as e:
            c.SendCommand("runGMMEstimation")
        assert e.value.code == -3                                  # incomplete configuration
        with pytest.raises(pocs.PocsError) as e:
            c.SendCommand("noSuchCommand 1")
        assert e.value.code == -5
        # everything configured except the collision world: the reference gets its scene through the
        # module constructor (mcsimplugin.cpp:12); a context that never got one must not answer 0.0
        c.configure(plan, dict(footprint=env["footprint"], boxes=None), K=3, N=100, seed=1)
        for cmd in ("runGMMEstimation", "runSimulation"):
            with pytest.raises(pocs.PocsError) as e:
                c.SendCommand(cmd)
            assert e.value.code == -3 and "collision world" in str(e.value)
        c.SendCommand("clearObstacles")                           # an explicitly empty world is a world
        assert float(c.SendCommand("runGMMEstimation")) == 0.0
        # positions inside a shard are 32-bit in the kernels: a larger shard is refused before anything is allocated
        c.SendCommand("setNumGMMSamples 3000000000")
        with pytest.raises(pocs.PocsError) as e:
            c.SendCommand("runGMMEstimation")
        assert e.value.code == -1 and "shard" in str(e.value)
        c.SendCommand("setNumGMMSamples 100")
        assert float(c.SendCommand("runSimulation")) == 0.0
        with pytest.raises(pocs.PocsError) as e:
            c.SendCommand("setNumGaussians 9")
        assert e.value.code == -1
        with pytest.raises(pocs.PocsError) as e:
            c.SendCommand("setQ abc")
        assert e.value.code == -1
        assert c.SendCommand("ArmaCommand") == ""
    finally:
        c.close()


def test_step_api_refuses_calls_out_of_sequence(pocs, plan, env):
    """The per-waypoint API of the sharded path: every call out of sequence is an error (POCS_E_ORDER),
    never a launch on half-built state -- the in-tail exchange included, which needs its buffers."""
    with pocs.Context(0) as c:
        c.configure(plan, env, K=3, N=2000, seed=3)
        for call in (lambda: c.gmm_sample_local(0), lambda: c.gmm_sample_exchange_local(0), lambda: c.gmm_end()):
            with pytest.raises(pocs.PocsError) as e:
                call()                                            # before gmm_begin
            assert e.value.code == -2
        c.gmm_begin()
        with pytest.raises(pocs.PocsError) as e:
            c.gmm_sample_exchange_local(0)                        # no exchange buffers connected
        assert e.value.code == -2 and "pocs_xchg_connect" in str(e.value)
        with pytest.raises(pocs.PocsError) as e:
            c.gmm_sample_local(0)                                 # the mixture of waypoint 0 does not exist yet
        assert e.value.code == -2
        c.gmm_advance_local(0)
        with pytest.raises(pocs.PocsError) as e:
            c.gmm_sample_local(1)                                 # waypoint 0 first
        assert e.value.code == -2
        c.xchg_connect([c.xchg_create(1, 0)])                     # a world of one: the exchange is with itself
        c.gmm_sample_exchange_local(0)
        with pytest.raises(pocs.PocsError) as e:
            c.gmm_end()                                           # 55 waypoints to go
        assert e.value.code == -2
        for w in range(1, 56):
            c.gmm_sample_exchange_local(w)
        p = c.gmm_end()
        c.set_seed(3)
        assert p == c.run_gmm_estimation()                        # and it is the whole-run call's result


def test_graphs_survive_buffer_growth(pocs, orc, plan, env):
    """A captured launch graph bakes device pointers in.  MC at batch 1 captures graph_mc; a GMM call
    with run-ahead 16 then grows the shared header / chain buffers (freed and reallocated); the next
    MC call must not replay the stale graph: counters bit-exact against the oracle before and after."""
    N = 3000
    cfg = orc.config(plan, env, K=3)
    with pocs.Context(0) as c:
        c.configure(plan, env, K=3, N=N, seed=77)
        n0 = c.mc_run_local()
        assert n0 == orc.run_mc(cfg, 77, N)[0]
        c.set_option(pocs.OPT_RUN_AHEAD, 16)
        c.set_seed(77)
        p = c.run_gmm_estimation()                              # R = 16: d_hdr / d_chain are replaced
        assert p == orc.run_gmm(cfg, 77, N)["prob"]
        c.set_option(pocs.OPT_RUN_AHEAD, 1)
        c.set_seed(77)
        assert c.mc_run_local() == n0
        _, hits = c.particles(N)
        assert np.array_equal(hits, orc.run_mc(cfg, 77, N)[1])


def test_rotated_obstacles_and_offset_footprint(ctx, orc, plan):
    """pr2custom.env.xml has boxes rotated by +-60 / 90 degrees: the primitive is a general OBB."""
    env2 = dict(footprint=[0.05, -0.02, 0.30, 0.25],
                boxes=np.array([[0.9, 1.0, 0.1, 0.6, math.radians(60)], [-2.0, -0.9, 0.5, 0.1, math.radians(-60)],
                                [1.8, -0.3, 0.2, 0.7, math.radians(90)], [0.0, 1.9, 4.0, 0.1, 0.0]]))
    cfg = orc.config(plan, env2, K=2)
    ctx.configure(plan, env2, K=2, N=6000, seed=3)
    p = ctx.run_gmm_estimation()
    want = orc.run_gmm(cfg, 3, 6000, want_samples=True)
    assert np.array_equal(ctx.gmm_samples(6000)[1], want["flags"])
    assert p == want["prob"]
    ctx.set_seed(3)
    assert ctx.run_simulation() == orc.run_mc(cfg, 3, 6000)[0] / 6000


def test_full_size_properties(ctx, pocs, plan, env):
    """BASELINE.json configs[2] at full size (500 waypoints, 10^7 samples, 8 components): the oracle
    would need ~25 min, so check size-independent properties instead."""
    big = pocs.resample_plan(plan, 500)
    N, K = 10_000_000, 8
    ctx.configure(big, env, K=K, N=N, seed=SEED)
    p1 = ctx.run_gmm_estimation()
    probs = ctx.waypoint_probabilities().copy()
    for w in (0, 17, 250, 499):
        m = ctx.moments(w, K)
        assert m[:, 0].sum() + m[:, 1].sum() == N            # every sample counted exactly once
        assert abs(m[:, 1].sum() / N - probs[w]) == 0.0
        _, covs, wts, _ = ctx.gmm_state(w, K)
        assert abs(wts.sum() - 1.0) < 1e-12
        for k in range(K):
            assert np.all(np.linalg.eigvalsh((covs[k] + covs[k].T) / 2) > 0)
    assert abs(p1 - (1.0 - np.prod(1.0 - probs))) < 1e-12      # F1: 1 - prod(1 - p_i)
    ctx.set_seed(SEED)
    assert ctx.run_gmm_estimation() == p1                       # bitwise reproducible
    assert 0.0 < p1 <= 1.0


def _snapshot(ctx, K, N, mc):
    """Everything a caller can read after a run."""
    if mc:
        xyz, hits = ctx.particles(N)
        return dict(xyz=xyz.copy(), hits=hits.copy(), counts=list(ctx.mc_batch_counts()),
                    bp=list(ctx.batch_probabilities()))
    xyz, flags = ctx.gmm_samples(N)
    hc = ctx.host_chain(8)
    return dict(probs=ctx.waypoint_probabilities().copy(), m7=ctx.moments(7, K).copy(), m55=ctx.moments(55, K).copy(),
                st=ctx.gmm_state_raw(30, K).copy(), xyz=xyz.copy(), flags=flags.copy(), app=hc["applied"].copy(),
                z=hc["z"].copy(), mu=hc["mu"].copy(), bp=list(ctx.batch_probabilities()))


def _same(a, b):
    return a.keys() == b.keys() and all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in a)


@pytest.mark.parametrize("mc", [False, True])
def test_run_ahead_serves_the_same_runs(pocs, plan, env, mc):
    """POCS_OPT_RUN_AHEAD: one run per call, the next R runs evaluated in one launch.  The sequence
    of results AND of everything the getters expose must be what one launch per run gives --
    across the refill (R = 4, 11 calls), a setter in the middle of a served group (the run counter
    resumes after the last run handed out) and a seed rewind."""
    K, N = 3, 4000

    def sequence(run_ahead):
        out = []
        with pocs.Context(0) as c:
            c.configure(plan, env, K=K, N=N, seed=41)
            c.set_num_particles(N)
            if run_ahead != 1:
                c.send_command("setRunAhead %d" % run_ahead)
            run = c.run_simulation if mc else c.run_gmm_estimation
            for i in range(11):
                out.append((run(), _snapshot(c, K, N, mc)))
            c.set_q(pocs.DEFAULTS["Q"])                      # a setter after run 2 of a group of 4
            for i in range(3):
                out.append((run(), _snapshot(c, K, N, mc)))
            c.set_seed(41)                                   # rewind: run 0 again
            out.append((run(), _snapshot(c, K, N, mc)))
            out.append((run(), _snapshot(c, K, N, mc)))
        return out

    one, ahead, auto = sequence(1), sequence(4), sequence(0)      # 0: the depth sized from N (64 here)
    assert len(one) == len(ahead) == len(auto) == 16
    for i, ((p1, _), (p3, s3)) in enumerate(zip(one, auto)):
        assert p1 == p3 and s3["bp"] == [p3], i
    for i, ((p1, s1), (p2, s2)) in enumerate(zip(one, ahead)):
        assert p1 == p2, i
        assert s2["bp"] == [p2], i                           # the caller asked for one run at a time
        assert _same(s1, s2), i                              # every getter, bit for bit: the launch shape changes nothing
    assert one[14][0] == one[0][0] and ahead[14][0] == ahead[0][0]               # after the rewind
    assert len({p for p, _ in one[:11]}) == 11                                   # every run redraws


def test_run_ahead_with_alternating_paths(pocs, plan, env):
    """GMM and MC calls interleaved: each call consumes one run index, whether or not the previous
    call left cached runs of the other path behind."""
    def sequence(run_ahead):
        with pocs.Context(0) as c:
            c.configure(plan, env, K=2, N=3000, seed=9)
            c.set_num_particles(3000)
            c.set_option(pocs.OPT_RUN_AHEAD, run_ahead)
            out = []
            for call in "GGMGMMMGG":
                out.append(c.run_gmm_estimation() if call == "G" else c.run_simulation())
            return out
    one, ahead = sequence(1), sequence(4)
    assert one == ahead


def test_mc_nontemporal_instantiation_matches(pocs, ctx, orc, plan, env):
    """k_mc_step<NT>: the instantiation the host selects once a batch's particle state exceeds the
    Infinity Cache (too big for this suite), forced here through POCS_OPT_MC_NONTEMPORAL: same hit
    counters and particles as the oracle, and as the plain instantiation."""
    N = 5001
    cfg = orc.config(plan, env, K=1)
    results = {}
    for nt in ("0", "1"):
        ctx.configure(plan, env, K=1, N=N, seed=SEED + 7)       # a setter: the launch graph is captured again
        ctx.set_option(pocs.OPT_MC_NONTEMPORAL, int(nt))
        ctx.set_num_particles(N)
        p = ctx.run_simulation()
        xyz, hits = ctx.particles(N)
        results[nt] = (p, xyz.copy(), hits.copy())
    n_mc, want_hits, _ = orc.run_mc(cfg, SEED + 7, N)
    ctx.set_option(pocs.OPT_MC_NONTEMPORAL, -1)
    for nt in ("0", "1"):
        assert results[nt][0] == n_mc / N and np.array_equal(results[nt][2], want_hits)
    assert np.array_equal(results["0"][1], results["1"][1])


def test_default_bench_shape_properties(pocs, plan, env):
    """The shape bench.py runs by default (64 runs of 10^6 samples per launch, K = 3), read back run
    by run through run-ahead: every sample counted exactly once at every waypoint, finite sums,
    live mixtures, F1 consistent, and the 64 runs all different."""
    N, K, R = 1_000_000, 3, 64
    with pocs.Context(0) as c:
        c.configure(plan, env, K=K, N=N, seed=SEED)
        c.set_option(pocs.OPT_RUN_AHEAD, R)
        ps = []
        for r in range(R):
            p = c.run_gmm_estimation()
            probs = c.waypoint_probabilities()
            assert abs(p - (1.0 - np.prod(1.0 - probs))) < 1e-12
            for w in (0, 31, 55):
                m = c.moments(w, K)
                assert np.all(np.isfinite(m)) and m[:, 0].sum() + m[:, 1].sum() == N
                assert abs(m[:, 1].sum() / N - probs[w]) == 0.0
            _, covs, wts, alive = c.gmm_state(55, K)
            assert abs(wts.sum() - 1.0) < 1e-12 and np.all(alive == 1.0)
            ps.append(p)
        assert len(set(ps)) == R and all(0.0 < p < 1.0 for p in ps)
        assert abs(np.mean(ps) - 0.2857) < 0.02          # this build's own level (profiles/r01_table1_like.txt, GMM3): a regression pin


def test_span_profile_times_the_replayed_graph(pocs, plan, env):
    """POCS_OPT_PROFILE = 2: one pair of events around the replayed graph of a whole-run call -- what bench.py divides
    by W for the launch period of the hot kernel.  The results do not change, the span covers W launches, and it is
    no longer than W launches bracketed one by one (POCS_OPT_PROFILE = 1, eager)."""
    K, N = 3, 200000
    with pocs.Context(0) as c:
        c.configure(plan, env, K=K, N=N, seed=SEED)
        c.set_batch(4)
        p0 = c.run_gmm_estimation()
        c.set_seed(SEED)
        c.set_option(pocs.OPT_PROFILE, 2)
        c.run_gmm_estimation()                                   # (captures the graph again: a setter was called)
        c.set_seed(SEED)
        p2 = c.run_gmm_estimation()
        ms2, n2 = c.kernel_time()
        c.set_seed(SEED)
        c.set_option(pocs.OPT_PROFILE, 1)
        p1 = c.run_gmm_estimation()
        ms1, n1 = c.kernel_time()
        c.set_seed(SEED)
        c.set_option(pocs.OPT_PROFILE, 0)
        c.set_num_particles(N)
        c.set_option(pocs.OPT_PROFILE, 2)
        c.run_simulation()
        c.set_seed(SEED)
        c.run_simulation()
        msm, nm = c.kernel_time()
        with pytest.raises(pocs.PocsError):
            c.set_option(pocs.OPT_PROFILE, 3)
    assert p0 == p1 == p2
    assert n2 == n1 == 56 and 0.0 < ms2 < 1.25 * ms1
    assert nm == 55 and msm > 0.0


def test_device_sampler_functions_on_edge_words(ctx, orc):
    """The device's table-driven Box-Muller pair and heading sine / cosine on inputs a free-running launch meets
    once in 2^32 draws, bit for bit against the oracle: the radius word 0 (the level u = 0: no logarithm; numerics v9
    gives it what the hardware's frexp and v_ffbh_u32 return for a zero word, and the oracle writes that case out),
    2^32 - 1, every power of two and its neighbours, the 512 cells' first and last words in several octaves; angle
    words on both sides of bit 23 and on the sector boundaries; headings on the sectors' ties, at zero, far out."""
    rng = np.random.default_rng(77)
    wr = [0, 1, 2, 3, 2 ** 32 - 1, 2 ** 32 - 2, 2 ** 31, 2 ** 31 - 1, 2 ** 31 + 1]
    wr += [(1 << e) + d for e in range(1, 32) for d in (-1, 0, 1)]
    wr += [(1 << e) + (i << max(e - 9, 0)) - d for e in (9, 10, 17, 24, 31) for i in range(0, 512, 7) for d in (0, 1)]
    wr += [int(v) for v in rng.integers(0, 2 ** 32, 4000)]
    wr = np.array([w & 0xFFFFFFFF for w in wr], dtype=np.uint32)
    n = len(wr)
    wa_edge = [0, 1, 2 ** 23 - 1, 2 ** 23, 2 ** 23 + 1, 2 ** 24 - 1, 2 ** 24, 2 ** 24 + 2 ** 23, 2 ** 31, 2 ** 32 - 1, 0x7F800000, 0x7F7FFFFF]
    wa = np.array((wa_edge * (n // len(wa_edge) + 1))[:n], dtype=np.uint64)
    wa[len(wa_edge) * 20:] = rng.integers(0, 2 ** 32, n - len(wa_edge) * 20)
    wa = wa.astype(np.uint32)
    x = np.concatenate([[0.0, -0.0, math.pi, -math.pi, 2 * math.pi, 1e-300, 123456.789, -99999.5],
                        (np.arange(-200, 200) + 0.5) * (math.pi / 128),                  # rint's ties
                        rng.uniform(-30, 30, n)])[:n]
    z0, z1, sn, cs, r2 = ctx.probe_device_math(wr, wa, x)
    for i in range(n):
        want = orc.normal_pair_w2(int(wr[i]), int(wa[i]))
        assert (z0[i], z1[i]) == want, (i, int(wr[i]), int(wa[i]), (z0[i], z1[i]), want)
        assert r2[i] == orc.radius2_unit32(int(wr[i])), (i, int(wr[i]))
        assert (sn[i], cs[i]) == orc.sincos_tab(float(x[i])), (i, x[i])
    assert 2.04 < math.sqrt(r2[0]) < 2.041 and np.all(np.isfinite(z0)) and np.all(np.isfinite(z1))
