/* pocs.h -- C ABI of libpocs.so, the MI355X-native collision-probability estimator.
 *
 * This is the drop-in boundary for the reference's OpenRAVE module `MCModule`
 * (mcsimplugin/mcsimplugin.cpp:7-255) and the estimator it owns (`MCSimulator`,
 * mcsimplugin/MCSimulator.h:93-930; `GM_Model`, mcsimplugin/GM_Model.h:34-126).
 * Plain C types only; no exceptions cross this boundary.  Every function that returns `int`
 * returns POCS_OK (0) or a negative POCS_E_* code; pocs_last_error() gives the text.
 *
 * One context = one GPU = one host thread at a time (the reference is single-threaded and takes
 * the environment mutex per query, MCSimulator.h:272,282).  Contexts are independent.
 * There is no CPU fallback: pocs_create fails when no HIP device is usable.
 */
#ifndef POCS_H
#define POCS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define POCS_OK 0
#define POCS_E_ARG (-1)        /* bad argument / wrong token count                           */
#define POCS_E_ORDER (-2)      /* command sent before the one it depends on (see below)      */
#define POCS_E_STATE (-3)      /* run requested with incomplete configuration                */
#define POCS_E_DEVICE (-4)     /* HIP error (text holds hipGetErrorString)                   */
#define POCS_E_UNKNOWN_COMMAND (-5)
#define POCS_E_BUFFER (-6)     /* caller buffer too small                                    */

typedef struct pocs_ctx pocs_ctx;

/* ---- lifetime: replaces CreateInterfaceValidated / MCModule ctor / DestroyPlugin ----------
 * (mcsimplugin.cpp:12-45,236-255; MCSimulator ctor MCSimulator.h:139-156). `device` is the HIP
 * ordinal. */
int pocs_create(pocs_ctx** out, int device);
void pocs_destroy(pocs_ctx* ctx);
const char* pocs_last_error(const pocs_ctx* ctx);
const char* pocs_version(void);

/* ---- collision world: replaces the OpenRAVE environment + robot handed to the MCSimulator
 * ctor and queried at MCSimulator.h:275,279.  boxes = M x {cx, cy, half_x, half_y, yaw_rad}. */
int pocs_set_footprint(pocs_ctx* ctx, double dx, double dy, double half_x, double half_y);
int pocs_set_obstacles(pocs_ctx* ctx, const double* boxes, int M);

/* ---- typed twins of the setter commands (argument order = token order of the command) ---- */
int pocs_set_alphas(pocs_ctx* ctx, const double* alphas, int n);            /* mcsimplugin.cpp:174-187 -> MCSimulator.h:224-230; n must be 4 */
int pocs_set_q(pocs_ctx* ctx, double q);                                    /* :168-172 -> :232-235 */
int pocs_set_num_landmarks(pocs_ctx* ctx, int n);                           /* :142-146 -> :208-212 */
int pocs_set_landmarks(pocs_ctx* ctx, const double* xs_then_ys, int n);     /* :148-166 -> :214-218; needs set_num_landmarks first */
int pocs_set_num_particles(pocs_ctx* ctx, long long n);                     /* :136-140 -> :182-186 */
int pocs_set_initial_covariance(pocs_ctx* ctx, const double* row_major9);   /* :121-134 -> :176-180 */
int pocs_set_path_length(pocs_ctx* ctx, int W);                             /* :115-119 -> :200-202 */
int pocs_set_trajectory(pocs_ctx* ctx, const double* xs_ys_thetas, int W);  /* :83-97 -> :158-168 (also sets the initial mean); needs set_path_length first */
int pocs_set_odometry(pocs_ctx* ctx, const double* r1s_trs_r2s, int Wm1);   /* :99-113 -> :170-174 */
int pocs_set_num_gaussians(pocs_ctx* ctx, int K);                           /* :49-54 -> :188-192; 1..8 */
int pocs_set_num_gmm_samples(pocs_ctx* ctx, long long n);                   /* :56-61 -> :194-198 */
int pocs_set_seed(pocs_ctx* ctx, uint64_t seed);                            /* new: the reference seeds from the clock (MCSimulator.h:141, GM_Model.h:53-54) */

/* ---- the two estimators ---------------------------------------------------------------- */
int pocs_run_simulation(pocs_ctx* ctx, double* probability);       /* runSimulation,    mcsimplugin.cpp:75-81 -> MCSimulator.h:361-365 */
int pocs_run_gmm_estimation(pocs_ctx* ctx, double* probability);   /* runGMMEstimation, mcsimplugin.cpp:66-72 -> MCSimulator.h:354-358 */

/* ---- the text channel: same 15 command names and token grammar as MCModule's SendCommand
 * (mcsimplugin.cpp:13-44) plus `setSeed`, `setFootprint`, `addObstacle`, `clearObstacles`,
 * `help`.  `line` = "<name> <tokens...>".  The reply text (the probability for run*, "output"
 * for MyCommand) is written NUL-terminated into out[cap]. */
int pocs_send_command(pocs_ctx* ctx, const char* line, char* out, size_t cap);

/* ---- options (ours) -------------------------------------------------------------------- */
#define POCS_OPT_STORE_SAMPLES 1   /* 1 (default): GMM samples + flags are written to HBM (26 B/eval, auditable); 0: not stored */
#define POCS_OPT_MC_FUSED 2        /* 0 (default): one launch per waypoint, particles streamed through HBM (56 B/eval); 1: whole roll-out in registers */
#define POCS_OPT_USE_GRAPH 3       /* 1 (default): the per-run launch sequence is replayed from a hipGraph */
#define POCS_OPT_PROFILE 4         /* 1: bracket every launch of the hot kernel with hipEvents (eager launches; see pocs_get_kernel_time).  An event
                                      between two kernels costs the bracketed kernel ~9 us of dispatch that back-to-back launches overlap.
                                      2 (whole-run GMM calls): the replayed graph as it runs in production between one pair of events;
                                      pocs_get_kernel_time then returns the span and W, i.e. the mean launch PERIOD (duration + gap) */
#define POCS_OPT_RUN_AHEAD 5       /* R > 1, or 0 = R sized per call from the sample / particle count (8..64: the reference's 200 runs
                                      of 10^4 samples go 64 at a time, a 10^6-sample estimation 16 at a time); default 1 = off.
                                      With one run per call (batch 1), a run* call evaluates the NEXT R
                                      runs of the context in one launch and the following R-1 calls are served from it -- the
                                      reference driver's one-command-per-run loop (MCSimulation.py:238-256) at batch throughput.
                                      Same runs, same seeds, and -- the moment sums being defined on a run's virtual slices, not on
                                      the launch -- bit for bit the same results and getters as one launch per run
                                      (tests/test_gpu_parity.py::test_timed_launch_shapes_against_the_oracle); any setter ends the
                                      serving and the run counter resumes after the last run handed out.  Text: setRunAhead R */
#define POCS_OPT_PERSISTENT 6      /* retired (round 3): 0 is accepted, 1 returns POCS_E_ARG.  Round 2's queue-driven whole-call kernel
                                      (k_gmm_run) was slower than one launch per waypoint at every batch size measured and is gone
                                      (DESIGN.md section 5). */
#define POCS_OPT_LONE_CALL 7       /* 1 (default): a whole-run GMM call of ONE run (batch 1, no run-ahead, one GPU) uses launches that close
                                      the previous waypoint in every block's head instead of tickets and a closing block (no one else
                                      is in flight to hide a closer behind): 10 % less time per waypoint, the same bits
                                      (tests/test_gpu_parity.py::test_lone_call_changes_no_bit).  0: the ticket form always. */
#define POCS_OPT_SUB_BATCHES 8     /* 0 (default): a whole-run call of >= 8 runs and >= 1.2e6 evaluations per waypoint is issued as TWO sub-batches
                                      on two streams, so that one sub-batch's launch tail (its last blocks' slow end, the serial mixture advance
                                      of its last closer, the launch boundary) is covered by the other's sampling blocks: +4 ... +11 % measured;
                                      smaller calls as one launch per waypoint.  1 / 2 force one form.  The moment sums are defined on a run's
                                      virtual slices, not on the launch: no bit of any result changes. */
#define POCS_OPT_MC_NONTEMPORAL 9  /* -1 (default): k_mc_step streams past the caches when the batch's particle state (28 B per particle)
                                      exceeds the 256 MB Infinity Cache and uses plain accesses when it fits; 0 / 1 force one form.
                                      Same results either way. */
int pocs_set_option(pocs_ctx* ctx, int option, long long value);

/* ---- batches of independent runs (ours) --------------------------------------------------
 * The reference's driver performs its 200 estimations one after the other
 * (MCSimulation.py:238-256).  With a batch of R, one pocs_run_gmm_estimation / begin..end
 * sequence advances R independent estimations in lockstep (one launch per waypoint for all of
 * them); run i of the batch draws exactly what the i-th of R consecutive single runs would have
 * drawn -- and computes, bit for bit, what it would have computed: the launch shape changes no result.
 * pocs_run_gmm_estimation returns run 0's probability, pocs_get_batch_probabilities all
 * R.  The per-waypoint exchange of the step API then covers R x 11K doubles.  runSimulation is
 * batched the same way (pocs_mc_get_batch_counts: the shard's collided particles per run). */
int pocs_set_batch(pocs_ctx* ctx, int runs);
int pocs_get_batch_probabilities(pocs_ctx* ctx, double* out, int cap);
int pocs_select_batch_run(pocs_ctx* ctx, int run);   /* the getters below (waypoint probabilities, moments, mixture state, host chain,
                                                        samples / particles) expose run `run` of the last batch; a new launch selects run 0 */

/* ---- sharding over GPUs (one process per GPU; the caller owns the collective) -----------
 * A context evaluates global sample / particle indices [first, first+count) of the N configured;
 * random draws are keyed by the GLOBAL index, so every sample, flag, survivor count and hit counter is the same
 * whatever the partition.  The moment SUMS of the GMM path are another matter: each shard adds its samples in its own
 * summation tree (a function of the shard's sample count: 512-pair chunks, at most 256 virtual slices -- part of the
 * numerics version pocs_version() names) and the ranks' totals are added in rank order, so the sums of a sharded run
 * differ from the one-GPU run's in their last bits (a different order of the same additions): bit-equality of sums,
 * mixture states and probabilities holds PER SHARD SIZE -- the same world size and shards reproduce bit for bit --,
 * not across world sizes (8 shards of cfg3 against one GPU: all counts equal, final probability equal, sums within
 * 8e-10 of their scale). */
int pocs_set_shard(pocs_ctx* ctx, long long first, long long count);   /* (-1, -1) = the whole range again */
int pocs_set_stream(pocs_ctx* ctx, void* hip_stream);                 /* launch on this stream (e.g. a torch stream).  NULL = back to the context's
                                                                          own NON-BLOCKING stream -- not the null stream: a caller whose
                                                                          other work (collectives, copies) runs on the null stream -- torch's
                                                                          default "current stream" has handle 0 -- passes hipStreamLegacy
                                                                          ((hipStream_t)1), or the two are not ordered (parallel.GpuEngine does) */

/* GMM, one waypoint at a time: begin -> for w in 0..W-1 { step_local(w); <all-reduce SUM of
 * moments_ptr(w), moments_len doubles>; } -> end.  step_local(w) first folds the (already
 * reduced) moments of w-1 into the mixture, then samples + collides + reduces this shard. */
int pocs_gmm_begin(pocs_ctx* ctx);
int pocs_gmm_step_local(pocs_ctx* ctx, int waypoint);
/* The two halves of step_local, for a caller that interleaves two contexts on one GPU and wants
 * the small advance launch to overlap the other context's sampling kernel: advance_local(w) builds
 * the mixture of waypoint w from the reduced moments of w-1, sample_local(w) samples + collides +
 * reduces; both on the context's stream, in this order. */
int pocs_gmm_advance_local(pocs_ctx* ctx, int waypoint);
int pocs_gmm_sample_local(pocs_ctx* ctx, int waypoint);
void* pocs_gmm_moments_ptr(pocs_ctx* ctx, int waypoint);              /* device pointer, f64[moments_len] */
int pocs_gmm_moments_len(const pocs_ctx* ctx);                       /* batch x 11 x K: one exchange covers every run of the batch */
int pocs_gmm_bind_moments(pocs_ctx* ctx, void* device_ptr, long long len_doubles);  /* optional: keep the [W][batch][11K] moments in a caller-owned device buffer (e.g. a torch tensor handed to all_reduce); NULL unbinds */
int pocs_gmm_end(pocs_ctx* ctx, double* probability);
/* The exchange without a collective library, for the GPUs of ONE node (11 K doubles per
 * run and waypoint are latency, not bandwidth): every rank owns a small device buffer that all ranks map
 * (HIP IPC); pocs_gmm_exchange_local(w) -- one small launch, after pocs_gmm_sample_local(w) -- writes this
 * rank's moments of waypoint w into its slot of EVERY rank's buffer (one hop over xGMI), waits for the
 * world's slots in its own buffer, adds them in rank order (every rank the same bits) and builds the
 * mixture of waypoint w+1, so the sequence per waypoint is sample_local(w), exchange_local(w) -- no
 * advance_local, no all-reduce.  Setup, once: pocs_xchg_create on every rank (returns the 64-byte IPC
 * handle of its buffer), the caller gathers the handles (any host channel), pocs_xchg_connect(handles of
 * rank 0 .. world-1, 64 bytes each).  world <= 8, batch <= 256.
 * Two rules for connected contexts: (1) LOCK STEP -- every rank calls pocs_gmm_begin the same number of times
 * and exchanges the same waypoints in the same order: the count of begin..end sequences is part of every row's
 * epoch and of the choice between the two slot sets; (2) NOBODY LEAVES EARLY -- a rank must not destroy its
 * context (or exit) while a peer may still write into or read from its buffer: put a barrier of the host
 * channel between the last pocs_gmm_end and pocs_destroy, as bench.py and the tests do.
 * A peer that never arrives makes the kernel's bounded wait (30 s, once per call) give up: pocs_gmm_end then
 * returns POCS_E_DEVICE and the call's results are discarded. */
int pocs_xchg_create(pocs_ctx* ctx, int world, int rank, void* handle64_out);
int pocs_xchg_connect(pocs_ctx* ctx, const void* handles_world_x_64, int world);
int pocs_gmm_exchange_local(pocs_ctx* ctx, int waypoint);
/* THE WHOLE CALL IN ONE LIBRARY CALL (round 4; what bench.py uses for N > 1 by default, POCS_ONEHOP=2, after a probe of it on
 * the node): a context that holds a shard (pocs_set_shard) and is connected to its peers (pocs_xchg_create / pocs_xchg_connect),
 * with no caller-owned moments buffer bound, runs the SHARDED estimation when pocs_run_gmm_estimation (or the text command) is
 * called on it: the library replays the call from its hipGraph -- the same launches, the same two sub-batches as on one GPU -- and
 * the block that closes a run's waypoint exchanges the run's moments with the other ranks over one hop, adds them in rank order
 * and builds the next mixture (as pocs_gmm_sample_exchange_local below does per waypoint).  Every rank returns the same
 * probabilities; the getters show the whole mixture's moments and states.  The call's number -- part of every row's epoch --
 * travels in the run headers uploaded per call, so the replayed graph needs no re-capture.  Rules as below: lock step (every
 * connected rank makes the same calls in the same order), nobody leaves early; a peer that never arrives makes the call return
 * POCS_E_DEVICE after the kernel's bounded wait.
 * The same exchange one waypoint at a time, launches issued by the caller (POCS_ONEHOP=3; round 3's default):
 * POCS_ONEHOP=1 is the two-launch form above, POCS_ONEHOP=0 one RCCL all-reduce per waypoint): the block that closes a run's
 * waypoint is also its messenger -- it sends the shard's moments, waits for the world's, adds them in rank
 * order and builds the next mixture, while the launch's finished blocks have already given their CUs to
 * whatever else is queued (a second context's sampling launch).  Per waypoint: sample_exchange_local(w);
 * before waypoint 0: advance_local(0).  Same results as sample_local + exchange_local, bit for bit. */
int pocs_gmm_sample_exchange_local(pocs_ctx* ctx, int waypoint);
/* MC: the shard's count of particles that collided at least once (device-synchronous). */
int pocs_mc_run_local(pocs_ctx* ctx, unsigned long long* collided);            /* run 0 of the batch */
int pocs_mc_get_batch_counts(pocs_ctx* ctx, unsigned long long* out, int cap);   /* every run of the last MC batch */

/* ---- results of the last run, for audits and parity tests ------------------------------- */
int pocs_get_path_length(const pocs_ctx* ctx);
int pocs_get_waypoint_probabilities(pocs_ctx* ctx, double* out, int cap);         /* `probabilities`, MCSimulator.h:660,678,817 */
int pocs_get_moments(pocs_ctx* ctx, int waypoint, double* out, int cap);          /* K x 11: nFree nColl Sx Sy St Sxx Sxy Sxt Syy Syt Stt */
int pocs_get_gmm_state(pocs_ctx* ctx, int waypoint, double* means3, double* covs9, double* weights, double* alive);  /* the mixture sampled at `waypoint`; alive[k] = 0 for a retired component */
int pocs_get_host_chain(pocs_ctx* ctx, double* applied3, double* noisy3, double* z, double* mu3, double* cov9); /* per step i<W-1; z is L per step */
long long pocs_copy_gmm_samples(pocs_ctx* ctx, double* xyt_aos, int16_t* flags, long long cap);  /* last waypoint's shard, 3 x n column-major like arma (x,y,theta triples) */
long long pocs_copy_particles(pocs_ctx* ctx, double* xyt_aos, uint32_t* hits, long long cap);   /* mcparticles / particlecollisions, MCSimulator.h:105,108 */
int pocs_measure_copy_bandwidth(pocs_ctx* ctx, long long bytes, double* gbps);  /* read+write GB/s of a plain streaming copy on this GPU: the measured HBM ceiling */
int pocs_measure_fill_bandwidth(pocs_ctx* ctx, long long bytes, double* gbps);  /* written GB/s of a plain streaming fill: the write-only ceiling (the GMM kernels read nothing) */
int pocs_get_kernel_time(pocs_ctx* ctx, double* total_ms, long long* launches);  /* hot-kernel time of the last run with POCS_OPT_PROFILE=1 (of the launches on the context's stream) */
int pocs_get_exchange_wait(pocs_ctx* ctx, double* min_median_max_us);  /* sharded GMM calls through the library's own exchange (pocs_gmm_exchange_local /
                                                                          pocs_gmm_sample_exchange_local): how long the closers of the last begin..end sequence
                                                                          waited for the other ranks' moments, over its (run, waypoint) pairs -- the first thing
                                                                          to read when a multi-GPU run scales badly (ranks that drift apart show here) */
int pocs_probe_device_math(pocs_ctx* ctx, int n, const uint32_t* radius_words, const uint32_t* angle_words, const double* headings,
                           double* z0, double* z1, double* sn, double* cs, double* radius2);
/* TEST HOOK, no counterpart in the reference: the DEVICE's table-driven sampler functions on inputs the
   caller picks -- per i < n the Box-Muller pair (z0, z1) of the words (radius_words[i], angle_words[i]) with its squared radius,
   and sin / cos of headings[i] as the footprint test evaluates them -- through the same inline functions, tables and pinned
   constants as k_gmm_step.  A free-running launch meets a given word once in 2^32 draws; this puts the edge words (0, the
   cells' boundaries, 2^32 - 1, a heading on a sector's tie) in front of the oracle directly (tests/test_gpu_parity.py). */
int pocs_get_sequence_time(pocs_ctx* ctx, double* ms, int* concurrent);  /* POCS_OPT_PROFILE=1, whole-run GMM calls: first sampling launch -> end of the last one, and how many
                                                                            sub-batches of the call were in flight side by side (their launches overlap: DESIGN.md section 5) */

#ifdef __cplusplus
}
#endif
#endif /* POCS_H */
