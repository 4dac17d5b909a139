"""Sharding of the two estimators over GPUs: one process per GPU, torch.distributed (RCCL).

The exchange protocol lives here, independent of what evaluates a shard, so that it is the same
code on 8 MI355X (engine = GpuEngine over libpocs.so) and in the CPU tests (world size 2, gloo,
an oracle-backed engine standing in for the kernels):

  MC   no data-path collective; ONE all_reduce(SUM) of the integer hit count at the end.
  GMM  one all_reduce(SUM) of the 11*K moment doubles per waypoint (MCSimulator.h:592-629 needs
       the global truncated moments before the next waypoint's mixture exists).

Random draws are keyed by the GLOBAL sample index, so the result does not depend on the split.
"""
import numpy as np

HIP_STREAM_LEGACY = 1          # hipStreamLegacy ((hipStream_t)1), hip_runtime_api.h: the null stream, by name


def shard_range(n_total, rank, world):
    """Contiguous, balanced split of range(n_total): returns (first, count) of `rank`.
    Shards start on even indices: mixture samples 2j and 2j+1 share their random draws."""
    n_total, world = int(n_total), int(world)
    pairs = (n_total + 1) // 2
    base, rem = divmod(pairs, world)
    p0 = rank * base + min(rank, rem)
    p1 = p0 + base + (1 if rank < rem else 0)
    lo, hi = min(2 * p0, n_total), min(2 * p1, n_total)
    return lo, hi - lo


def combine_probabilities(colliding_per_waypoint, n_total):
    """p_w = colliding / N (MCSimulator.h:633-641); result = 1 - prod(1 - p_w) (:848-856)."""
    probs = np.asarray(colliding_per_waypoint, dtype=np.float64) / float(n_total)
    prod = 1.0
    for p in probs:
        prod *= (1.0 - p)
    return 1.0 - prod, probs


def run_gmm_sharded(engine, dist=None):
    """engine: begin(), step_local(w), moments(w) -> tensor view [11*K] (device of the engine),
    end() -> probability, attributes W.  dist: torch.distributed (initialised) or None."""
    engine.begin()
    for w in range(engine.W):
        engine.step_local(w)
        if dist is not None:
            dist.all_reduce(engine.moments(w))
    return engine.end()


def run_gmm_onehop(engines):
    """The same waypoint loop with the exchange done by the library itself (pocs_gmm_exchange_local:
    every rank writes its moments into its slot of every rank's IPC-mapped buffer -- one hop over
    xGMI --, adds the slots of its own buffer in rank order and builds the next mixture, all in one
    small launch): per waypoint sample(w), exchange(w); no collective, no host in the loop.
    Engines must have been connected (GpuEngine.connect_onehop).  Two engines alternate as in
    run_gmm_pipelined."""
    for e in engines:
        with e.stream_ctx():
            e.begin()
            e.advance(0)
    prev = None
    for w in range(engines[0].W):
        for e in engines:
            with e.stream_ctx():
                if prev is not None and len(engines) > 1:
                    e.wait_event(prev)
                e.sample(w)
                prev = e.record_event()
                e.exchange(w)
    out = []
    for e in engines:
        with e.stream_ctx():
            out.append(e.end())
    return out


def run_gmm_onehop_fused(engines):
    """run_gmm_onehop with ONE launch per waypoint and engine (pocs_gmm_sample_exchange_local): the block
    that closes a run's waypoint exchanges its moments and advances the mixture in the sampling launch's
    tail.  Nothing orders the two engines' launches except, once, the start: the second engine's first
    launch waits for the first engine's -- from then on each engine's next launch becomes ready while the
    other's blocks hold the CUs and takes them as they finish, so an engine's exchange (its closers wait
    for the other ranks) always runs beside the other engine's sampling."""
    first = None
    for e in engines:
        with e.stream_ctx():
            e.begin()
            e.advance(0)
    for w in range(engines[0].W):
        for i, e in enumerate(engines):
            with e.stream_ctx():
                if w == 0 and i > 0 and first is not None:
                    e.wait_event(first)
                e.sample_exchange(w)
                if w == 0 and i == 0:
                    first = e.record_event()
    out = []
    for e in engines:
        with e.stream_ctx():
            out.append(e.end())
    return out


def run_gmm_pipelined(engines, dist):
    """Two (or more) engines, each with its own batch of runs and its own stream, advanced waypoint
    by waypoint in turn: while one engine's moments are in the all-reduce (and its small mixture
    advance runs), the other engine's sampling kernel has the GPU.  The sampling kernels alternate
    strictly -- each waits for the previous engine's sampling kernel, an event, before it starts --
    because two of them in flight would share the CUs block by block and reach their tails
    together.  Collectives are issued from this one thread in the same order on every rank.
    Returns the list of per-engine probabilities."""
    for e in engines:
        with e.stream_ctx():
            e.begin()
    prev = None                                      # "the previous sampling kernel has finished"
    for w in range(engines[0].W):
        for e in engines:
            with e.stream_ctx():
                e.advance(w)                         # overlaps the other engine's sampling kernel
                if prev is not None and len(engines) > 1:
                    e.wait_event(prev)
                e.sample(w)
                prev = e.record_event()
                if dist is not None:
                    dist.all_reduce(e.moments(w))
    out = []
    for e in engines:
        with e.stream_ctx():
            out.append(e.end())
    return out


def connect_contexts(ctx, dist, rank, world):
    """A context with a shard set (pocs_set_shard) joins its peers: every rank creates its exchange buffer, the 64-byte IPC
    handles go round once over the host channel of the process group, every rank maps every buffer.  From then on
    ctx.run_gmm_estimation() IS the sharded estimation: the library replays the whole call from a graph -- the same two
    sub-batches as on one GPU -- and every run's moments cross the ranks over one hop in the closing block of its launch;
    every rank returns the same probabilities.  Connected contexts make the same calls in the same order (include/pocs.h)."""
    mine = ctx.xchg_create(world, rank)
    if dist is not None and world > 1:
        handles = [None] * world
        dist.all_gather_object(handles, mine)
    else:
        handles = [mine]
    ctx.xchg_connect(handles)


def run_mc_sharded(engine, n_total, dist=None):
    """engine: mc_local() -> int64 tensor with the shard's collided count of every run of the
    batch.  Returns run 0's probability (all of them: engine.last_mc_probabilities)."""
    cnt = engine.mc_local()
    if dist is not None:
        dist.all_reduce(cnt)
    probs = [int(v) / float(n_total) for v in cnt.tolist()]
    engine.last_mc_probabilities = probs
    return probs[0]


class GpuEngine:
    """A libpocs context bound to this rank's GPU, torch's current stream and a torch-owned
    moments buffer (so all_reduce can take views of it)."""

    def __init__(self, ctx, W, K, n_total, rank=0, world=1, per_rank=None, batch=1, stream=None):
        import torch
        self.ctx, self.W, self.K, self.torch = ctx, W, K, torch
        self.stream = stream                   # a torch.cuda.Stream of its own (pipelined engines) or None
        self.max_batch = batch
        self.batch = batch                     # independent runs advanced in lockstep (pocs_set_batch)
        ctx.set_batch(batch)
        first, count = (rank * per_rank, per_rank) if per_rank else shard_range(n_total, rank, world)
        ctx.set_shard(first, count)
        self.count = count
        # launch on a torch stream so kernels and collectives are ordered without host syncs.  torch's default
        # "current stream" is the null stream, whose handle is 0 -- which pocs_set_stream reads as "the context's
        # own (non-blocking) stream": name the null stream by hipStreamLegacy instead, or the caller's collectives
        # on the current stream would not be ordered with the context's launches at all.
        handle = (stream or torch.cuda.current_stream()).cuda_stream
        ctx.set_stream(handle if handle else HIP_STREAM_LEGACY)
        self.buf = torch.zeros(W * batch * K * 11, dtype=torch.float64, device="cuda")   # [W][batch][K*11]
        ctx.gmm_bind_moments(self.buf.data_ptr(), self.buf.numel())
        torch.cuda.synchronize()               # the zero fill ran on torch's default stream

    def stream_ctx(self):
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def set_batch(self, b):
        """Fewer runs than the engine was sized for (last, partial call of a sequence)."""
        assert 1 <= b <= self.max_batch
        self.ctx.set_batch(b)
        self.batch = b

    def begin(self):
        self.ctx.gmm_begin()

    def step_local(self, w):
        self.ctx.gmm_step_local(w)

    def advance(self, w):
        self.ctx.gmm_advance_local(w)

    def sample(self, w):
        self.ctx.gmm_sample_local(w)

    def record_event(self):
        ev = self.torch.cuda.Event()
        ev.record(self.torch.cuda.current_stream())
        return ev

    def wait_event(self, ev):
        self.torch.cuda.current_stream().wait_event(ev)

    def connect_onehop(self, dist, rank, world):
        """Set up the library's own exchange: every rank creates its buffer, the 64-byte IPC handles go
        round once over the host channel of the process group, every rank maps every buffer."""
        mine = self.ctx.xchg_create(world, rank)
        if dist is not None and world > 1:
            handles = [None] * world
            dist.all_gather_object(handles, mine)
        else:
            handles = [mine]
        self.ctx.xchg_connect(handles)

    def exchange(self, w):
        self.ctx.gmm_exchange_local(w)

    def sample_exchange(self, w):
        self.ctx.gmm_sample_exchange_local(w)

    def moments(self, w):
        n = self.batch * self.K * 11           # one exchange per waypoint covers every run of the batch
        return self.buf[w * n:(w + 1) * n]

    def end(self):
        return self.ctx.gmm_end()

    def probabilities(self):
        return self.ctx.batch_probabilities()

    def mc_local(self):
        self.ctx.mc_run_local()
        return self.torch.tensor(self.ctx.mc_batch_counts(), dtype=self.torch.int64, device="cuda")
