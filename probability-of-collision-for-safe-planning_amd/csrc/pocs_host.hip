// pocs_host.hip -- host runtime behind the C ABI (include/pocs.h): configuration state, the
// per-waypoint host chain, device buffers, launch sequences (eager or replayed from a hipGraph)
// and the MCModule-compatible text dispatcher.
//
// Mirrors, on the host side: MCModule (mcsimplugin/mcsimplugin.cpp:7-232) and the O(1) part of
// MCSimulator::EKF_GaussProp (mcsimplugin/MCSimulator.h:649-864).  Everything per particle /
// per sample runs in pocs_kernels.hip.  There is no CPU path for that work: without a HIP device
// pocs_create fails.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>
#include <map>
#include <chrono>
#include <algorithm>

#include "../../include/pocs.h"
#include "pocs_kernels.h"
#include "pocs_command.hpp"

#define POCS_VERSION_STRING "pocs-mi355x 0.4 (gfx950; numerics v9: summation tree of 512-pair chunks, 256 virtual slices)"

namespace {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct pocs_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t own_stream = nullptr;
  // A call of many runs is issued as `groups` sub-batches on streams of their own (gmm_groups): while one
  // sub-batch is in the tail of a waypoint's launch (the last blocks' slower waves, the serial mixture
  // advance of its closers, the launch boundary) the others' sampling blocks have the SIMDs.
  hipStream_t side_stream[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr}, ev_seq[2] = {nullptr, nullptr};
  double seq_ms = 0.0;                   // POCS_OPT_PROFILE: first launch -> last launch's end of the last whole-run call
  int seq_groups = 1;
  std::string err;

  // ---- configuration (what MCSimulator holds, MCSimulator.h:94-136) ----
  double alphas[4] = {1, 1, 1, 1};      // ones, as the reference ctor leaves them (:143)
  bool have_alphas = false;
  pocs_sensor sensor;
  bool have_q = false, have_landmarks = false;
  int num_landmarks = -1;
  long long num_particles = -1;
  double cov0[9];
  bool have_cov0 = false;
  int W = -1;
  std::vector<double> traj, odom;        // by component: 3 x W, 3 x (W-1)
  bool have_traj = false, have_odom = false;
  int K = -1;
  long long num_gmm = -1;
  uint64_t seed = 0x5EED0001ull;
  uint64_t run_index = 0;
  pocs_footprint fp = {0.0, 0.0, 0.334, 0.334};
  std::vector<double> boxes;             // M x 5
  bool have_obstacles = false;           // pocs_set_obstacles / addObstacle / clearObstacles was called at least once
  long long shard_first = -1, shard_count = -1;
  long long opt_store = 1, opt_fused = 0, opt_graph = 1, opt_profile = 0, opt_lone = 1, opt_groups = 0, opt_mc_nt = -1;
  unsigned long long epoch = 0;          // bumped by every setter; part of the graph cache key
  int batch = 1;                         // independent GMM estimations advanced in lockstep per call
  // run-ahead (POCS_OPT_RUN_AHEAD): with batch == 1 a run* call evaluates the next `run_ahead` runs
  // of the context in one launch and the following calls are served from it; `view` is the run of
  // the last launch the getters expose.
  int run_ahead = 1;
  int view = 0;
  int ra_have = 0;                       // runs of the last launch that may still be served (0: none)
  int ra_kind = 0;                       // 1 GMM, 2 MC
  int last_kind = 0;                     // what the last launch was: 1 GMM, 2 MC
  bool ra_internal = false;              // the last launch was an internal run-ahead batch
  uint64_t batch_base = 0;               // run_index of run 0 of the last launch
  int batch_R = 1;                       // runs in the last launch
  std::vector<double> batch_moments;     // [W][R][K*11] of the last GMM launch

  // host image (headers | chains | initial mixtures) of the NEXT batch, computed while the GPU
  // works on the current one
  struct {
    bool valid = false;
    uint64_t seed = 0, run_index = 0;
    int R = 0;
    unsigned long long epoch = 0;
    std::vector<double> image, chain0, mu0, cov0;
  } ahead;

  // ---- device state ----
  DevBuf d_env, d_sensor, d_hdr, d_chain, d_state, d_param, d_moments, d_partial;
  DevBuf d_sx, d_sy, d_st, d_flags, d_px, d_py, d_pt, d_hits, d_total, d_ticket, d_tables;
  // one-hop exchange (pocs_xchg_*): this rank's buffer, the peers' buffers as mapped here
  void* xchg_own = nullptr;
  void* xchg_peer[POCS_XCHG_MAX_WORLD] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int xchg_world = 0, xchg_rank = -1;
  bool xchg_connected = false;
  unsigned long long xchg_calls = 0;     // begin/end sequences so far: part of every row's epoch
  double* ext_moments = nullptr;         // caller-owned moments buffer (multi-GPU), or null
  long long ext_moments_len = 0;
  void* h_pin = nullptr;                 // pinned staging: hdr | chain | state0 | moments | total
  size_t h_pin_cap = 0;
  void* h_copy = nullptr;                // pinned staging of the audit copies (pocs_copy_*, pocs_get_gmm_state): device data
                                         // reaches caller memory through it, in pieces of POCS_COPY_CHUNK bytes
  bool env_dirty = true, sensor_dirty = true;

  hipGraphExec_t graph_gmm = nullptr, graph_mc = nullptr;
  const void* graph_baked[3] = {nullptr, nullptr, nullptr};   // diagnostic build only (POCS_GRAPH_WITH_COPIES)
  std::string graph_gmm_key, graph_mc_key;

  std::vector<hipEvent_t> events;
  double prof_ms = 0.0;
  long long prof_launches = 0;

  // ---- results of the last run ----
  std::vector<double> h_chain;           // (W-1) x POCS_CHAIN_STRIDE
  std::vector<double> h_mu, h_cov;       // (W-1) x 3, (W-1) x 9 : main EKF after each step
  std::vector<double> probs;             // W (run 0 of the last batch)
  std::vector<double> batch_probs;       // final probability of every run of the last batch
  std::vector<unsigned long long> mc_counts;   // collided particles of every run of the last MC batch (this shard)
  std::vector<double> last_moments;      // W x K x 11
  long long last_gmm_count = 0, last_mc_count = 0;
  int last_gmm_wp = -1;
  int last_gmm_adv = -1;                 // last waypoint whose mixture has been built (step API)
  bool gmm_open = false;
};

namespace {

int fail(pocs_ctx* c, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

#define HIPCHK(c, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess)                                                                 \
      return fail((c), POCS_E_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                  __FILE__, __LINE__);                                                    \
  } while (0)

void drop_graphs(pocs_ctx* c) {
  if (c->graph_gmm) { hipGraphExecDestroy(c->graph_gmm); c->graph_gmm = nullptr; }
  if (c->graph_mc) { hipGraphExecDestroy(c->graph_mc); c->graph_mc = nullptr; }
  c->graph_gmm_key.clear();
  c->graph_mc_key.clear();
}

// Grow a device buffer.  The captured graphs bake device pointers in (d_hdr and d_chain are shared
// by the GMM and the MC graph), so replacing ANY buffer drops both of them: the next run captures
// again against the new pointers.
int ensure(pocs_ctx* c, DevBuf& b, size_t bytes) {
  if (bytes == 0) bytes = 16;
  if (b.cap >= bytes) return POCS_OK;
  if (b.p) {
    HIPCHK(c, hipStreamSynchronize(c->stream));      // nothing queued may still use the old buffer
    drop_graphs(c);
    HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.cap = 0;
  }
  HIPCHK(c, hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return POCS_OK;
}

int grid_blocks(long long count, int block, int bpc) {
  // One block per `block` evaluations up to `bpc` resident blocks per CU (256 CUs), grid-stride beyond that.
  long long nb = (count + block - 1) / block;
  if (nb < 1) nb = 1;
  if (nb > 256LL * bpc) nb = 256LL * bpc;
  if (nb > POCS_MAX_BLOCKS) nb = POCS_MAX_BLOCKS;
  return (int)nb;
}
// Task geometry of k_gmm_step (pocs_kernels.h): a chunk = one block iteration = TB pairs of samples; a
// run's `chunks` are cut into VS = 2^vs_shift virtual slices -- a function of the shard's sample count
// ONLY, the moment sums are defined on them -- and the launch's units (run, virtual slice) are dealt to
// the blocks `upb` at a time.  The chip takes blocks 256 at a time (one more per CU) and holds 512 of
// these: `upb` is the smallest number that fits the launch into 512 blocks (256 below 4 runs: fewer,
// fatter blocks amortise head and tail better when the launch is short anyway), so every block of a launch
// has the same amount of work whatever the number of runs (20 runs x 256 slices = 512 blocks x 10).
#define POCS_FULL_GRID_FROM_RUNS 4   // (round 4, measured after the heads got shorter: 512 blocks from 4 runs per launch on, +4 % at 4 runs, +6-7 % at 6 and 7; 2 and 3 runs lose 4 % with them -- profiles/r04_ab_full_grid_from_4_runs.txt; it was 8)
struct GmmGeometry { long long chunks; int vs_shift; int upb; int blocks; };
GmmGeometry gmm_geometry(long long count, int runs, int K, int groups = 1) {
  const int tb = POCS_GMM_BLOCK_OF(K);
  const long long npairs = (count + 1) / 2;
  GmmGeometry g;
  g.chunks = (npairs + tb - 1) / tb;
  if (g.chunks < 1) g.chunks = 1;
  g.vs_shift = 0;
  while ((2LL << g.vs_shift) <= g.chunks && (2 << g.vs_shift) <= POCS_GMM_MAX_VS) ++g.vs_shift;
  if (runs < 1) runs = 1;
  const long long units = (long long)runs << g.vs_shift;
  // (`groups` launches share the chip: each gets its share of the resident blocks)
  const long long budget = (runs * groups >= POCS_FULL_GRID_FROM_RUNS ? POCS_NUM_CUS * POCS_GMM_BLOCKS_PER_CU : POCS_NUM_CUS) / groups;
  g.upb = (int)((units + budget - 1) / budget);
  if (g.upb > (1 << g.vs_shift)) g.upb = 1 << g.vs_shift;           // a block's range touches at most two runs
  g.blocks = (int)((units + g.upb - 1) / g.upb);
  return g;
}
int grid_for_mc(long long count, int runs = 1) {                                      // MC kernels, per run
  // six 256-thread blocks per CU (69 VGPRs: plenty of room) keep enough loads in flight for the
  // stream: measured 0.78 of the HBM peak against 0.67 with three, 0.70 with eight
  const int one = grid_blocks(count, POCS_BLOCK, 6);
  if (runs <= 1) return one;
  int per = 1536 / runs;                             // the batch's launches share the chip's 6 x 256 resident blocks
  if (per < 1) per = 1;
  return per < one ? per : one;
}

// ---------------------------------------------------------------------------------------------
// Host chain: everything in EKF_GaussProp's loop body that does not touch particles/samples
// (MCSimulator.h:692-800): M (generateM_EKF :495-513), gain L and applied control (:714-726,
// generateL :532-553, inverseOdometry :434-449), EKFpredict on the main estimate (:746),
// sampleOdometry (:754, :391-410), the L noisy range observations (:786-789, :383-387) and
// EKFupdate (:797-800).  Noise comes from Philox stream POCS_STREAM_CHAIN, index = step,
// draw order r1, tr, r2, z_0 .. z_{L-1} as in the reference (:403-405, :786-789).
// ---------------------------------------------------------------------------------------------
void inverse_odometry(const double p1[3], const double p2[3], double out[3]) {
  double r1 = atan2(p2[1] - p1[1], p2[0] - p1[0]) - p1[2];
  r1 = pocs_wrap_angle(r1);
  const double ddx = p2[0] - p1[0], ddy = p2[1] - p1[1];
  const double tr = sqrt(ddx * ddx + ddy * ddy);
  double r2 = p2[2] - p1[2] - r1;
  r2 = pocs_wrap_angle(r2);
  out[0] = r1; out[1] = tr; out[2] = r2;
}

double chain_normal(uint64_t seed, int step, int draw) {
  const pocs_u32x4 w = pocs_draw(seed, (uint64_t)step, 0u, POCS_STREAM_CHAIN, (uint32_t)(draw >> 1));
  double n0, n1;
  pocs_normal_pair(w.x, w.y, w.z, &n0, &n1);
  return (draw & 1) ? n1 : n0;
}

void compute_chain(pocs_ctx* c, uint64_t seed) {
  const int W = c->W, L = c->sensor.L;
  c->h_chain.assign((size_t)(W > 1 ? W - 1 : 1) * POCS_CHAIN_STRIDE, 0.0);
  c->h_mu.assign((size_t)(W > 1 ? W - 1 : 1) * 3, 0.0);
  c->h_cov.assign((size_t)(W > 1 ? W - 1 : 1) * 9, 0.0);
  double mu[3] = {c->traj[0], c->traj[W], c->traj[2 * W]};
  double cov[9];
  memcpy(cov, c->cov0, sizeof cov);
  double real[3] = {mu[0], mu[1], mu[2]};
  const double a1 = c->alphas[0], a2 = c->alphas[1], a3 = c->alphas[2], a4 = c->alphas[3];
  for (int i = 0; i < W - 1; ++i) {
    double* rec = &c->h_chain[(size_t)i * POCS_CHAIN_STRIDE];
    const double us[3] = {c->odom[i], c->odom[(W - 1) + i], c->odom[2 * (W - 1) + i]};
    const double xs[3] = {c->traj[i], c->traj[W + i], c->traj[2 * W + i]};
    const double xg[3] = {c->traj[i + 1], c->traj[W + i + 1], c->traj[2 * W + i + 1]};
    // generateM_EKF on the NOMINAL control
    rec[3] = a1 * (us[0] * us[0]) + a2 * (us[1] * us[1]);
    rec[4] = a3 * (us[1] * us[1]) + a4 * (us[0] * us[0]) + a4 * (us[2] * us[2]);
    rec[5] = a1 * (us[2] * us[2]) + a2 * (us[1] * us[1]);
    // generateL + applied control
    double ureq[3], applied[3];
    inverse_odometry(mu, xg, ureq);
    for (int j = 0; j < 3; ++j) {
      const double xhat = mu[j] - xs[j];
      const double ubar = ureq[j] - us[j];
      const double gain = ubar / (xhat != 0 ? xhat : 0.1);
      applied[j] = us[j] + gain * xhat;
      rec[j] = applied[j];
    }
    // EKFpredict on the main estimate
    double pmu[3], pcov[9];
    pocs_ekf_predict(mu, cov, applied, rec + 3, pmu, pcov);
    // sampleOdometry on the APPLIED control
    const double v0 = a1 * (applied[0] * applied[0]) + a2 * (applied[1] * applied[1]);
    const double v1 = a3 * (applied[1] * applied[1]) +
                      a4 * ((applied[0] * applied[0]) + (applied[2] * applied[2]));
    const double v2 = a1 * (applied[2] * applied[2]) + a2 * (applied[1] * applied[1]);
    double noisy[3];
    noisy[0] = applied[0] + chain_normal(seed, i, 0) * sqrt(v0);
    noisy[1] = applied[1] + chain_normal(seed, i, 1) * sqrt(v1);
    noisy[2] = applied[2] + chain_normal(seed, i, 2) * sqrt(v2);
    rec[6] = noisy[0]; rec[7] = noisy[1]; rec[8] = noisy[2];
    double next[3];
    pocs_motion(real, noisy, next);
    real[0] = next[0]; real[1] = next[1]; real[2] = next[2];
    // noisy range observations of the real state
    for (int l = 0; l < L; ++l) {
      const double dx = real[0] - c->sensor.lx[l], dy = real[1] - c->sensor.ly[l];
      const double dist = sqrt(dx * dx + dy * dy);
      rec[POCS_CHAIN_Z + l] = dist + (0.0 + chain_normal(seed, i, 3 + l) * sqrt(c->sensor.Q));
    }
    pocs_ekf_update(pmu, pcov, rec + POCS_CHAIN_Z, &c->sensor);
    memcpy(mu, pmu, sizeof mu);
    memcpy(cov, pcov, sizeof cov);
    memcpy(&c->h_mu[(size_t)i * 3], mu, sizeof mu);
    memcpy(&c->h_cov[(size_t)i * 9], cov, sizeof cov);
  }
}

int check_common(pocs_ctx* c) {
  if (!c->have_q || !c->have_landmarks) return fail(c, POCS_E_STATE, "setQ / setLandmarks missing");
  if (!c->have_cov0) return fail(c, POCS_E_STATE, "setInitialCovariance missing");
  if (!c->have_traj || !c->have_odom) return fail(c, POCS_E_STATE, "setTrajectory / setOdometry missing");
  if (c->W < 1) return fail(c, POCS_E_STATE, "setPathLength missing");
  // The reference receives its collision world through the module constructor (sim(penv),
  // mcsimplugin.cpp:12 -> MCSimulator.h:139-156).  A context that was never given one would answer
  // 0.0 for every run: refuse instead.  An explicitly empty world (clearObstacles) is allowed.
  if (!c->have_obstacles)
    return fail(c, POCS_E_STATE, "no collision world: pocs_set_obstacles / addObstacle / clearObstacles missing");
  return POCS_OK;
}

int upload_tables(pocs_ctx* c) {               // log / sector tables of the numerics spec, once
  if (c->d_tables.p) return POCS_OK;
  pocs_tables T;
  pocs_tables_init(&T);
  if (int r = ensure(c, c->d_tables, sizeof T)) return r;
  HIPCHK(c, hipMemcpy(c->d_tables.p, &T, sizeof T, hipMemcpyHostToDevice));
  return POCS_OK;
}
int upload_static(pocs_ctx* c) {
  if (int r = upload_tables(c)) return r;
  if (c->env_dirty) {
    pocs_env_dev env;
    memset(&env, 0, sizeof env);
    env.fp = c->fp;
    env.M = (int)(c->boxes.size() / 5);
    for (int m = 0; m < env.M; ++m)
      pocs_prepare_obstacle(&c->boxes[(size_t)m * 5], &c->fp, &env.obs[(size_t)m * POCS_OBS_STRIDE]);
    if (int r = ensure(c, c->d_env, sizeof env)) return r;
    HIPCHK(c, hipMemcpy(c->d_env.p, &env, sizeof env, hipMemcpyHostToDevice));
    c->env_dirty = false;
  }
  if (c->sensor_dirty) {
    if (int r = ensure(c, c->d_sensor, sizeof(pocs_sensor))) return r;
    HIPCHK(c, hipMemcpy(c->d_sensor.p, &c->sensor, sizeof(pocs_sensor), hipMemcpyHostToDevice));
    c->sensor_dirty = false;
  }
  return POCS_OK;
}

// pinned staging layout (doubles): [0 .. 2R) run headers, then R chains, then R initial mixtures,
// then the moments [W][R][K*11], then the MC total
struct PinLayout { size_t chain, state0, moments, total, end; };
PinLayout pin_layout(const pocs_ctx* c) {
  PinLayout p;
  const size_t W = (size_t)(c->W > 0 ? c->W : 1), K = (size_t)(c->K > 0 ? c->K : 1), R = (size_t)c->batch;
  p.chain = 2 * R;
  p.state0 = p.chain + R * (W > 1 ? W - 1 : 1) * POCS_CHAIN_STRIDE;
  p.moments = p.state0 + R * K * POCS_STATE_STRIDE;
  p.total = p.moments + W * R * K * POCS_NMOM;
  p.end = p.total + R + 2;                 // one u64 per run: MC totals
  return p;
}

int ensure_pin(pocs_ctx* c) {
  const size_t bytes = pin_layout(c).end * sizeof(double);
  if (c->h_pin_cap >= bytes) return POCS_OK;
  if (c->h_pin) { HIPCHK(c, hipHostFree(c->h_pin)); c->h_pin = nullptr; c->h_pin_cap = 0; }
  HIPCHK(c, hipHostMalloc(&c->h_pin, bytes, hipHostMallocDefault));
  c->h_pin_cap = bytes;
  drop_graphs(c);   // captured copies hold the old staging pointers
  return POCS_OK;
}

uint64_t effective_seed(const pocs_ctx* c, uint64_t ahead = 0) {
  // every run of a context draws a fresh stream (the reference re-draws on each run*,
  // MCSimulator.h:656-679); setSeed rewinds run_index so (seed, run) is reproducible.
  return c->seed + 0x9E3779B97F4A7C15ull * (c->run_index + ahead);
}

uint64_t seed_of_run(const pocs_ctx* c, uint64_t run) { return c->seed + 0x9E3779B97F4A7C15ull * run; }

// Run-ahead bookkeeping.  A setter (or any other launch) ends the serving of cached runs: the
// context's run counter goes back to just after the last run that was handed out, so the sequence
// of seeds the caller sees is the one it would have seen one run per launch.
void ra_drop(pocs_ctx* c) {
  if (c->ra_have > 0) c->run_index = c->batch_base + (uint64_t)c->view + 1;
  c->ra_have = 0;
}
void touch(pocs_ctx* c) { ra_drop(c); c->epoch++; }

double* moments_dev(pocs_ctx* c) { return c->ext_moments ? c->ext_moments : (double*)c->d_moments.p; }

int prof_begin(pocs_ctx* c, size_t launches) {
  c->prof_ms = 0.0; c->prof_launches = 0;
  if (c->opt_profile != 1) return POCS_OK;
  while (c->events.size() < 2 * launches) {
    hipEvent_t e;
    HIPCHK(c, hipEventCreate(&e));
    c->events.push_back(e);
  }
  return POCS_OK;
}
int prof_collect(pocs_ctx* c, size_t launches) {
  if (c->opt_profile != 1) return POCS_OK;
  for (size_t i = 0; i < launches; ++i) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->events[2 * i], c->events[2 * i + 1]));
    c->prof_ms += ms;
  }
  c->prof_launches = (long long)launches;
  return POCS_OK;
}

// ------------------------------------------------------------------------------------------
// GMM path
// ------------------------------------------------------------------------------------------
int gmm_shard(pocs_ctx* c, long long* first, long long* count) {
  *first = c->shard_first >= 0 ? c->shard_first : 0;
  *count = c->shard_first >= 0 ? c->shard_count : c->num_gmm;
  if (*first < 0 || *count < 0 || *first + *count > c->num_gmm)
    return fail(c, POCS_E_ARG, "shard [%lld,+%lld) outside numGMMSamples=%lld", *first, *count, c->num_gmm);
  if (*count > 2147480000LL)        // the kernels' positions inside a shard are 32-bit
    return fail(c, POCS_E_ARG, "a GMM shard holds at most 2147480000 samples per run (got %lld): shard the run", *count);
  if ((*first & 1) && *count > 0)   // mixture samples 2j, 2j+1 share their random draws: a shard starts on a pair
    return fail(c, POCS_E_ARG, "GMM shard must start at an even sample index (got %lld)", *first);
  return POCS_OK;
}

// The synchronisation words of one call (pocs_kernels.h): [1] give-up code, [0], [2..3] pad, then the
// tickets [R][W], then -- sharded runs -- the closers' exchange waits [R][W]; a block of its own, a multiple of
// 16 bytes, zeroed by ONE memset at the head of every call.
size_t sync_ticket_offset(const pocs_ctx*) { return 4; }
size_t sync_xwait_offset(const pocs_ctx* c) { return sync_ticket_offset(c) + (size_t)c->batch * (size_t)(c->W > 0 ? c->W : 1); }
size_t sync_words(const pocs_ctx* c) {
  const size_t n = sync_xwait_offset(c) + (size_t)c->batch * (size_t)(c->W > 0 ? c->W : 1);      // tickets, then the exchange waits
  return (n + 3) & ~(size_t)3;
}

long long sample_stride_of(long long count) { return count > 0 ? ((count + 1) & ~1LL) : 2; }   // even

int gmm_prepare(pocs_ctx* c) {
  if (int r = check_common(c)) return r;
  if (c->K < 1) return fail(c, POCS_E_STATE, "setNumGaussians missing");
  if (c->num_gmm < 1) return fail(c, POCS_E_STATE, "setNumGMMSamples missing");
  long long first, count;
  if (int r = gmm_shard(c, &first, &count)) return r;
  if (int r = upload_static(c)) return r;
  const size_t W = (size_t)c->W, K = (size_t)c->K, R = (size_t)c->batch;
  const GmmGeometry geo = gmm_geometry(count, c->batch, c->K);
  if (int r = ensure(c, c->d_hdr, R * sizeof(pocs_run_header))) return r;
  if (int r = ensure(c, c->d_chain, R * (W > 1 ? W - 1 : 1) * POCS_CHAIN_STRIDE * sizeof(double))) return r;
  if (int r = ensure(c, c->d_state, R * W * K * POCS_STATE_STRIDE * sizeof(double))) return r;
  if (int r = ensure(c, c->d_param, R * W * K * POCS_PARAM_STRIDE * sizeof(double))) return r;
  if (!c->ext_moments)
    if (int r = ensure(c, c->d_moments, W * R * K * POCS_NMOM * sizeof(double))) return r;
  if (c->ext_moments && c->ext_moments_len < (long long)(W * R * K * POCS_NMOM))
    return fail(c, POCS_E_BUFFER, "bound moments buffer too small");
  if (int r = ensure(c, c->d_partial, 2 * (R << geo.vs_shift) * K * POCS_NMOM * sizeof(double))) return r;   // (x 2: a lone call alternates halves)
  if (int r = ensure(c, c->d_ticket, sync_words(c) * sizeof(unsigned))) return r;
  if (c->opt_store) {
    const size_t n = R * (size_t)sample_stride_of(count);
    if (int r = ensure(c, c->d_sx, n * sizeof(double))) return r;
    if (int r = ensure(c, c->d_sy, n * sizeof(double))) return r;
    if (int r = ensure(c, c->d_st, n * sizeof(double))) return r;
    if (int r = ensure(c, c->d_flags, n * sizeof(int16_t))) return r;
  }
  return ensure_pin(c);
}

// host staging -> device: run header, chain, initial mixture (initGMM, MCSimulator.h:350-352,
// GM_Model.h:57-77: K copies of (mu0, Sigma0), weights 1/K)
// Host image of one batch starting at run `base` (relative to c->run_index): per run the header
// (seed), the chain record and the initial mixture (initGMM, MCSimulator.h:350-352,
// GM_Model.h:57-77: K copies of (mu0, Sigma0), weights 1/K), laid out as the pinned staging area.
// Leaves run 0's chain in c->h_chain / h_mu / h_cov.
void build_run_image(pocs_ctx* c, uint64_t base, double* img) {
  const PinLayout pl = pin_layout(c);
  const int W = c->W, R = c->batch;
  const size_t steps = (size_t)(W > 1 ? W - 1 : 1);
  for (int r = R - 1; r >= 0; --r) {          // run 0 last: c->h_chain / h_mu / h_cov keep ITS chain
    const uint64_t seed = effective_seed(c, base + (uint64_t)r);
    compute_chain(c, seed);
    pocs_run_header hdr; hdr.seed = seed; hdr.pad = c->xchg_calls;      // (sharded whole calls read their exchange epoch from here)
    memcpy(img + 2 * (size_t)r, &hdr, sizeof hdr);
    memcpy(img + pl.chain + (size_t)r * steps * POCS_CHAIN_STRIDE, c->h_chain.data(), c->h_chain.size() * sizeof(double));
    for (int k = 0; k < c->K; ++k) {
      double* s = img + pl.state0 + ((size_t)r * c->K + k) * POCS_STATE_STRIDE;
      s[0] = c->traj[0]; s[1] = c->traj[W]; s[2] = c->traj[2 * W];
      memcpy(s + 3, c->cov0, 9 * sizeof(double));
      s[12] = 1.0 / c->K; s[13] = 1.0; s[14] = 0.0; s[15] = 0.0;
    }
  }
}

// While the GPU works on the current batch: the host chains of the next one.
void prefetch_next_batch(pocs_ctx* c) {
  const PinLayout pl = pin_layout(c);
  auto& a = c->ahead;
  std::vector<double> keep_chain = c->h_chain, keep_mu = c->h_mu, keep_cov = c->h_cov;
  a.image.resize(pl.moments);
  build_run_image(c, 0, a.image.data());       // c->run_index already points at the next batch
  a.chain0.swap(c->h_chain); a.mu0.swap(c->h_mu); a.cov0.swap(c->h_cov);
  c->h_chain.swap(keep_chain); c->h_mu.swap(keep_mu); c->h_cov.swap(keep_cov);
  a.seed = c->seed; a.run_index = c->run_index; a.R = c->batch; a.epoch = c->epoch;
  a.valid = true;
}

// Host image of this call's batch into the pinned staging area (from the look-ahead cache when it
// matches), run counter advanced; then the uploads every path needs: headers and chains.
int stage_and_upload_runs(pocs_ctx* c) {
  const PinLayout pl = pin_layout(c);
  double* pin = (double*)c->h_pin;
  const int W = c->W, R = c->batch;
  const size_t steps = (size_t)(W > 1 ? W - 1 : 1);
  auto& a = c->ahead;
  if (a.valid && a.seed == c->seed && a.run_index == c->run_index && a.R == R && a.epoch == c->epoch &&
      a.image.size() == pl.moments) {
    memcpy(pin, a.image.data(), pl.moments * sizeof(double));
    c->h_chain = a.chain0; c->h_mu = a.mu0; c->h_cov = a.cov0;
  } else {
    build_run_image(c, 0, pin);
  }
  a.valid = false;
  for (int r = 0; r < R; ++r) ((uint64_t*)pin)[2 * (size_t)r + 1] = c->xchg_calls;      // (the image may have been built a call ago: the headers' exchange count is this call's)
  c->batch_base = c->run_index;
  c->batch_R = R;
  c->view = 0;
  c->run_index += (uint64_t)R;
  HIPCHK(c, hipMemcpyAsync(c->d_hdr.p, pin, (size_t)R * sizeof(pocs_run_header), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->d_chain.p, pin + pl.chain, (size_t)R * steps * POCS_CHAIN_STRIDE * sizeof(double),
                           hipMemcpyHostToDevice, c->stream));
  return POCS_OK;
}

int gmm_upload_run(pocs_ctx* c) {
  if (int r = stage_and_upload_runs(c)) return r;
  const PinLayout pl = pin_layout(c);
  double* pin = (double*)c->h_pin;
  const int W = c->W, R = c->batch;
  // the initial mixture of run r goes to state[r][0]: R rows of K*16 doubles, pitch W*K*16
  const size_t row = (size_t)c->K * POCS_STATE_STRIDE * sizeof(double);
  HIPCHK(c, hipMemcpy2DAsync(c->d_state.p, (size_t)W * row, pin + pl.state0, row, row, (size_t)R,
                             hipMemcpyHostToDevice, c->stream));
  return POCS_OK;
}

// A whole-run call of a context that is CONNECTED to its peers (pocs_xchg_connect) and holds a shard exchanges every
// run's moments in the closing block of its launch (k_gmm_step, exchange_in_tail): the call is then the sharded
// estimation in ONE library call -- the same graph replay, the same two sub-batches as on one GPU, no host in the loop.
bool whole_call_exchanges(const pocs_ctx* c) { return c->xchg_connected && c->shard_first >= 0 && !c->ext_moments; }

void fill_gmm_launch(pocs_ctx* c, pocs_gmm_launch* a, long long first, long long count, int w,
                     int run_lo = 0, int run_cnt = -1, int groups = 1) {
  memset(a, 0, sizeof *a);
  if (run_cnt < 0) run_cnt = c->batch;
  a->hdr = (const pocs_run_header*)c->d_hdr.p;
  a->env = (const pocs_env_dev*)c->d_env.p;
  a->tables = (const pocs_tables*)c->d_tables.p;
  a->chain = (const double*)c->d_chain.p;
  a->sensor = (const pocs_sensor*)c->d_sensor.p;
  a->state = (double*)c->d_state.p;
  a->param = (double*)c->d_param.p;
  a->moments = moments_dev(c);
  a->partial = (double*)c->d_partial.p;
  a->partial_prev = a->partial;
  a->sync = (unsigned*)c->d_ticket.p;
  a->ticket = a->sync + sync_ticket_offset(c);
  a->xwait = a->sync + sync_xwait_offset(c);
  const GmmGeometry geo = gmm_geometry(count, run_cnt, c->K, groups);
  a->chunks = geo.chunks; a->vs_shift = geo.vs_shift; a->upb = geo.upb; a->blocks = geo.blocks;
  a->run_lo = run_lo; a->run_cnt = run_cnt;
  a->x = (double*)c->d_sx.p; a->y = (double*)c->d_sy.p; a->th = (double*)c->d_st.p;
  a->flags = (int16_t*)c->d_flags.p;
  a->first = first; a->count = count; a->n_total = c->num_gmm;
  a->fp = c->fp; a->M = (int)(c->boxes.size() / 5);
  a->fp_rr = sqrt(c->fp.hx * c->fp.hx + c->fp.hy * c->fp.hy); a->fp_phi = atan2(c->fp.hy, c->fp.hx);
  a->waypoint = w; a->store = c->opt_store ? 1 : 0;
  a->sample_stride = sample_stride_of(count);
  a->nruns = c->batch; a->W = c->W;
}

// state/param[w] from state/moments[w-1]: its own tiny launch for waypoint 0 and, when sharded,
// after the caller's all-reduce; on one GPU the last block of k_gmm_step(w-1) has done it already.
int enqueue_advance(pocs_ctx* c, int w) {
  pocs_gmm_launch a;
  fill_gmm_launch(c, &a, 0, 0, w);
  HIPCHK(c, pocs_launch_gmm_advance(c->K, a, c->stream));
  return POCS_OK;
}

// One run per call (no batch, no run-ahead) on one GPU: the launches close the previous waypoint in their heads
// (k_gmm_step, "LONE"): 30.6 -> 27.5 us per waypoint at 10^6 samples, K = 3 (MI355X).  POCS_OPT_LONE_CALL = 0
// keeps the ticket-and-closer form; the results are the same bits.
bool lone_call(const pocs_ctx* c) { return c->opt_lone && c->batch == 1 && !c->ext_moments && !(c->xchg_connected && c->shard_first >= 0); }
void set_lone(pocs_ctx* c, pocs_gmm_launch* a, int w) {
  const size_t half = ((size_t)1 << a->vs_shift) * c->K * POCS_NMOM;      // one run's rows
  a->lone = 1;
  a->advance_in_tail = 0;
  a->partial = (double*)c->d_partial.p + (size_t)(w & 1) * half;
  a->partial_prev = (double*)c->d_partial.p + (size_t)((w + 1) & 1) * half;
}

int enqueue_step(pocs_ctx* c, long long first, long long count, int w, bool advance_in_tail, int prof_slot,
                 hipStream_t stream = nullptr, int run_lo = 0, int run_cnt = -1, int groups = 1, bool lone = false) {
  pocs_gmm_launch a;
  if (!stream) stream = c->stream;
  fill_gmm_launch(c, &a, first, count, w, run_lo, run_cnt, groups);
  a.advance_in_tail = (advance_in_tail && w + 1 < c->W) ? 1 : 0;
  if (lone) set_lone(c, &a, w);
  if (whole_call_exchanges(c)) {
    a.exchange_in_tail = 1;
    a.xchg_epoch_from_header = 1;
    for (int q = 0; q < c->xchg_world; ++q) a.xchg.buf[q] = (double*)c->xchg_peer[q];
    a.xchg.world = c->xchg_world; a.xchg.rank = c->xchg_rank;
  }
  if (prof_slot >= 0) HIPCHK(c, hipEventRecord(c->events[2 * prof_slot], stream));
  HIPCHK(c, pocs_launch_gmm_step(c->K, a, stream));
  if (prof_slot >= 0) HIPCHK(c, hipEventRecord(c->events[2 * prof_slot + 1], stream));
  return POCS_OK;
}

int enqueue_ticket_reset(pocs_ctx* c) {
  HIPCHK(c, hipMemsetAsync(c->d_ticket.p, 0, sync_words(c) * sizeof(unsigned), c->stream));
  return POCS_OK;
}

// How many launches of the hot kernel one whole-run call makes (what POCS_OPT_PROFILE brackets).
size_t gmm_hot_launches(const pocs_ctx* c) { return (size_t)c->W; }

// Sub-batches of a whole-run call (above).  The moment sums do not depend on the launch shape (pocs_kernels.hip,
// "summation tree"), so a split changes no bit of any result.  TWO sub-batches on two streams by default where a call
// has the work for it (round 4, measured on MI355X with numerics v8, 10^6 samples, K = 3, one box, three alternations and
// a sweep, profiles/r04_sub_batches.txt): while one sub-batch is in the tail of its waypoint -- the slow end of its last
// blocks, the serial mixture advance of its last closer, the launch boundary, the next launch's heads -- the other's
// sampling blocks have the chip: +11 % at 8 runs per call, +7 % at 16 and 20, +6 % at 32, +4 % at 64; K = 8, 10^7
// samples, 16 runs: +4 %.  Below 8 runs per call a launch of half the runs does not fill the chip (-5 ... -23 % at 2 ... 6
// runs), and launches of less than ~6 x 10^5 evaluations are all dispatch (64 runs of 10^4 samples: -15 %): one launch
// for all then.  Three sub-batches lose everywhere, four gain less than two.  (Round 3 had measured +0.5 % at 20 runs
// for the same split and left it off: its closers were a third longer and it timed one pass, not a median.)
// POCS_OPT_SUB_BATCHES: 0 = this rule (default), 1, 2 (tests/test_gpu_parity.py checks the bits).
int gmm_groups(const pocs_ctx* c) {
  if (c->ext_moments) return 1;                      // the caller's all-reduce covers the whole batch at once
  int g = (int)c->opt_groups;
  if (g == 0) {
    const double count = (double)(c->shard_first >= 0 ? c->shard_count : c->num_gmm);
    g = (c->batch >= 8 && (double)c->batch * count >= 1.2e6) ? 2 : 1;
  }
  if (g > c->batch) g = c->batch;
  return g < 1 ? 1 : g;
}

// The launches of a whole-run call: what the hipGraph holds.  KERNEL NODES ONLY -- the ticket reset ahead of
// them and the result copies behind them are plain stream operations (enqueue_gmm_results).  On ROCm 7.2 a
// captured graph that also held the memset and the two device-to-host copies went stale between replays: after
// a few dozen small synchronous copies plus a large one on the null stream (a caller reading mixture states
// and samples back between two runs), the next replay's memset no longer cleared the tickets and its kernels
// ran on garbage -- reproduced with round 2's library, gone with kernel-only graphs (tests/test_gpu_parity.py
// ::test_graph_replays_survive_readbacks).
int enqueue_gmm_all(pocs_ctx* c, long long first, long long count, bool prof) {
  const int W = c->W, R = c->batch, G = gmm_groups(c);
  if (int r = enqueue_advance(c, 0)) return r;
  if (prof) HIPCHK(c, hipEventRecord(c->ev_seq[0], c->stream));
  if (G > 1) HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
  for (int g = 1; g < G; ++g) HIPCHK(c, hipStreamWaitEvent(c->side_stream[g - 1], c->ev_fork, 0));
  const bool lone = lone_call(c);
  for (int w = 0; w < W; ++w)
    for (int g = 0; g < G; ++g) {                    // sub-batch g = runs [g R / G, (g + 1) R / G); events bracket sub-batch 0's launches
      const int lo = (int)((long long)g * R / G), hi = (int)((long long)(g + 1) * R / G);
      if (int r = enqueue_step(c, first, count, w, true, (prof && g == 0) ? w : -1, g == 0 ? c->stream : c->side_stream[g - 1], lo, hi - lo, G, lone)) return r;
    }
  if (lone) {                                        // the last waypoint's rows -> moments[W-1]
    pocs_gmm_launch a;
    fill_gmm_launch(c, &a, first, count, W - 1, 0, 1, 1);
    set_lone(c, &a, W - 1);
    HIPCHK(c, pocs_launch_gmm_close(c->K, a, c->stream));
  }
  for (int g = 1; g < G; ++g) {
    HIPCHK(c, hipEventRecord(c->ev_join[g - 1], c->side_stream[g - 1]));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[g - 1], 0));
  }
  if (prof) HIPCHK(c, hipEventRecord(c->ev_seq[1], c->stream));
  return POCS_OK;
}
int enqueue_gmm_results(pocs_ctx* c) {
  const int W = c->W;
  const PinLayout pl = pin_layout(c);
  HIPCHK(c, hipMemcpyAsync((double*)c->h_pin + pl.moments, moments_dev(c),
                           (size_t)W * c->batch * c->K * POCS_NMOM * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  // the call's give-up word travels back with the results
  HIPCHK(c, hipMemcpyAsync((double*)c->h_pin + pl.total + c->batch + 1, (unsigned*)c->d_ticket.p + POCS_SYNC_ABORT,
                           sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  return POCS_OK;
}

// F1 (MCSimulator.h:848-856): p_w = colliding / numGMMSamples (:633-641), result = 1 - prod(1 - p_w)
// The getters' view of the last GMM launch: per-waypoint probabilities and moments of run v.
void gmm_select_view(pocs_ctx* c, int v) {
  const int W = c->W, K = c->K, R = c->batch_R;
  c->view = v;
  c->probs.assign(W, 0.0);
  c->last_moments.assign((size_t)W * K * POCS_NMOM, 0.0);
  for (int w = 0; w < W; ++w) {
    const double* m = &c->batch_moments[((size_t)w * R + v) * K * POCS_NMOM];
    double coll = 0.0;
    for (int k = 0; k < K; ++k) coll += m[(size_t)k * POCS_NMOM + 1];
    c->probs[w] = coll / (1.0 * (double)c->num_gmm);
    memcpy(&c->last_moments[(size_t)w * K * POCS_NMOM], m, (size_t)K * POCS_NMOM * sizeof(double));
  }
}

void gmm_combine(pocs_ctx* c, const double* moments, double* probability) {
  const int W = c->W, K = c->K, R = c->batch;          // moments: [W][R][K*11]
  c->batch_R = R;
  c->last_kind = 1;
  c->batch_moments.assign(moments, moments + (size_t)W * R * K * POCS_NMOM);
  c->batch_probs.assign(R, 0.0);
  for (int r = 0; r < R; ++r) {
    double prod = 1.0;
    for (int w = 0; w < W; ++w) {
      const double* m = moments + ((size_t)w * R + r) * K * POCS_NMOM;
      double coll = 0.0;
      for (int k = 0; k < K; ++k) coll += m[(size_t)k * POCS_NMOM + 1];
      const double p = coll / (1.0 * (double)c->num_gmm);
      prod *= (1.0 - p);
    }
    c->batch_probs[r] = 1.0 - prod;
  }
  gmm_select_view(c, 0);
  *probability = c->batch_probs[0];
}

std::string config_key(const pocs_ctx* c, long long first, long long count, const char* tag) {
  char buf[256];
  snprintf(buf, sizeof buf, "%s e%llu W%d K%d R%d g%d l%d x%d n%lld f%lld c%lld s%lld fu%lld st%p em%p", tag, c->epoch,
           c->W, c->K, c->batch, gmm_groups(c), lone_call(c) ? 1 : 0, (c->xchg_connected && c->shard_first >= 0 && !c->ext_moments) ? c->xchg_world : 0,
           c->num_gmm, first, count, c->opt_store, c->opt_fused, (void*)c->stream, (void*)c->ext_moments);
  return buf;
}

int run_gmm_full(pocs_ctx* c, double* probability) {
  if (!probability) return fail(c, POCS_E_ARG, "null output");
#if defined(POCS_TUNING) && defined(POCS_CALL_TIMES)                          // tuning build: host-side phases of a call on stderr
  const auto t_in = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) { fprintf(stderr, "[call] %s +%.1f us\n", what,
      std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_in).count()); };
#else
  auto lap = [](const char*) {};
#endif
  if (int r = gmm_prepare(c)) return r;
  long long first, count;
  if (int r = gmm_shard(c, &first, &count)) return r;
  lap("prepared");
  if (whole_call_exchanges(c)) {
    if (c->batch > POCS_XCHG_MAX_RUNS) return fail(c, POCS_E_ARG, "exchange: at most %d runs per call", POCS_XCHG_MAX_RUNS);
    c->xchg_calls += 1;                                // one exchange sequence: every connected rank makes the same calls in the same order
  }
  if (int r = gmm_upload_run(c)) return r;
  lap("upload enqueued");
  // POCS_OPT_PROFILE: 1 = events around every launch of the hot kernel (eager launches: an event between two kernels
  // keeps the command processor from preparing the next launch under the running one, which adds ~9 us to what it
  // brackets); 2 = the replayed graph as it runs in production between ONE pair of events outside it: span / W is the
  // mean launch PERIOD -- duration plus the gap to the next launch --, an upper bound of the mean duration.
  const bool prof = c->opt_profile == 1, span = c->opt_profile == 2 && c->opt_graph;
  if (int r = prof_begin(c, gmm_hot_launches(c))) return r;
#if defined(POCS_TUNING) && defined(POCS_GRAPH_WITH_COPIES)      // diagnostic build: round 2's graph shape (the memset and the two result copies as graph nodes)
  const bool copies_in_graph = c->opt_graph && !prof;
#else
  const bool copies_in_graph = false;
#endif
  if (!copies_in_graph) if (int r = enqueue_ticket_reset(c)) return r;
  if (c->opt_graph && !prof) {
    const std::string key = config_key(c, first, count, "gmm");
    if (!c->graph_gmm || key != c->graph_gmm_key) {
      if (c->graph_gmm) { hipGraphExecDestroy(c->graph_gmm); c->graph_gmm = nullptr; }
      hipGraph_t g = nullptr;
      HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      int r = copies_in_graph ? enqueue_ticket_reset(c) : POCS_OK;
      if (!r) r = enqueue_gmm_all(c, first, count, false);
      if (!r && copies_in_graph) r = enqueue_gmm_results(c);
      if (copies_in_graph) {                           // the pointers the memset / memcpy nodes bake in: a stale one would show here
        c->graph_baked[0] = c->d_ticket.p; c->graph_baked[1] = c->h_pin; c->graph_baked[2] = moments_dev(c);
      }
      hipError_t e = hipStreamEndCapture(c->stream, &g);
      if (r) { if (g) hipGraphDestroy(g); return r; }
      HIPCHK(c, e);
      e = hipGraphInstantiate(&c->graph_gmm, g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      HIPCHK(c, e);
      c->graph_gmm_key = key;
    }
    if (copies_in_graph && (c->graph_baked[0] != c->d_ticket.p || c->graph_baked[1] != c->h_pin || c->graph_baked[2] != moments_dev(c)))
      return fail(c, POCS_E_STATE, "a pointer baked into the graph's memset / memcpy nodes changed between capture and replay");
    if (span) HIPCHK(c, hipEventRecord(c->ev_seq[0], c->stream));
    HIPCHK(c, hipGraphLaunch(c->graph_gmm, c->stream));
    if (span) HIPCHK(c, hipEventRecord(c->ev_seq[1], c->stream));
  } else {
    if (int r = enqueue_gmm_all(c, first, count, prof)) return r;
  }
  if (!copies_in_graph) if (int r = enqueue_gmm_results(c)) return r;
  lap("launched");
  prefetch_next_batch(c);          // host chains of the next batch, while the GPU works on this one
  lap("next batch prepared");
  HIPCHK(c, hipStreamSynchronize(c->stream));
  lap("synchronised");
  if (int r = prof_collect(c, gmm_hot_launches(c))) return r;
  if (prof || span) {
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev_seq[0], c->ev_seq[1]));
    c->seq_ms = ms; c->seq_groups = gmm_groups(c);
    if (span) { c->prof_ms = ms; c->prof_launches = (long long)gmm_hot_launches(c); }      // pocs_get_kernel_time: span / W
  }
  {
    unsigned gave_up = 0;
    memcpy(&gave_up, (double*)c->h_pin + pin_layout(c).total + c->batch + 1, sizeof gave_up);
    if (gave_up) return fail(c, POCS_E_DEVICE, "a bounded wait expired on the device (code %u); results discarded", gave_up);
  }
  gmm_combine(c, (double*)c->h_pin + pin_layout(c).moments, probability);
  c->last_gmm_count = count;
  c->last_gmm_wp = c->W - 1;
  lap("combined");
  return POCS_OK;
}

// ------------------------------------------------------------------------------------------
// MC path
// ------------------------------------------------------------------------------------------
int mc_shard(pocs_ctx* c, long long* first, long long* count) {
  *first = c->shard_first >= 0 ? c->shard_first : 0;
  *count = c->shard_first >= 0 ? c->shard_count : c->num_particles;
  if (*first < 0 || *count < 0 || *first + *count > c->num_particles)
    return fail(c, POCS_E_ARG, "shard [%lld,+%lld) outside numParticles=%lld", *first, *count, c->num_particles);
  return POCS_OK;
}

int enqueue_mc_all(pocs_ctx* c, long long first, long long count, bool prof) {
  const int W = c->W, R = c->batch, nblk = grid_for_mc(count, R);
  pocs_mc_launch a;
  memset(&a, 0, sizeof a);
  a.hdr = (const pocs_run_header*)c->d_hdr.p;
  a.env = (const pocs_env_dev*)c->d_env.p;
  a.tables = (const pocs_tables*)c->d_tables.p;
  a.chain = (const double*)c->d_chain.p;
  a.x = (double*)c->d_px.p; a.y = (double*)c->d_py.p; a.th = (double*)c->d_pt.p;
  a.hits = (uint32_t*)c->d_hits.p;
  a.total = (unsigned long long*)c->d_total.p;
  a.first = first; a.count = count; a.stride = sample_stride_of(count);
  a.W = W; a.nruns = R;
  // 28 B of state per particle.  Up to 8 x 10^6 particles (224 MB) the state of a batch stays in the
  // 256 MB Infinity Cache between waypoint launches; past that the launches stream from HBM whatever
  // they do, and non-temporal accesses then stream faster (16 x 10^6: 148 us instead of 189 us)
  a.nontemporal = c->opt_mc_nt >= 0 ? (int)c->opt_mc_nt : (((double)R * (double)a.stride * 28.0 > 232.0e6) ? 1 : 0);
  a.mu0[0] = c->traj[0]; a.mu0[1] = c->traj[W]; a.mu0[2] = c->traj[2 * W];
  if (!pocs_chol3_lower(c->cov0, a.L0)) return fail(c, POCS_E_ARG, "initial covariance is not positive definite");
  if (c->opt_fused) {
    a.step = W - 1;
    if (prof) HIPCHK(c, hipEventRecord(c->events[0], c->stream));
    HIPCHK(c, pocs_launch_mc_fused(nblk, a, c->stream));
    if (prof) HIPCHK(c, hipEventRecord(c->events[1], c->stream));
  } else {
    a.step = 0;
    HIPCHK(c, pocs_launch_mc_init(nblk, a, c->stream));
    for (int s = 0; s < W - 1; ++s) {
      a.step = s;
      if (prof) HIPCHK(c, hipEventRecord(c->events[2 * s], c->stream));
      HIPCHK(c, pocs_launch_mc_step(nblk, a, c->stream));
      if (prof) HIPCHK(c, hipEventRecord(c->events[2 * s + 1], c->stream));
    }
  }
  HIPCHK(c, pocs_launch_mc_count(nblk, a, c->stream));
  return POCS_OK;
}

// One batch of MC roll-outs (runSimulation x batch) over this context's shard; fills c->mc_counts.
int run_mc_local(pocs_ctx* c) {
  if (int r = check_common(c)) return r;
  if (c->num_particles < 1) return fail(c, POCS_E_STATE, "setNumParticles missing");
  long long first, count;
  if (int r = mc_shard(c, &first, &count)) return r;
  if (int r = upload_static(c)) return r;
  const size_t W = (size_t)c->W, R = (size_t)c->batch, n = R * (size_t)sample_stride_of(count);
  if (int r = ensure(c, c->d_hdr, R * sizeof(pocs_run_header))) return r;
  if (int r = ensure(c, c->d_chain, R * (W > 1 ? W - 1 : 1) * POCS_CHAIN_STRIDE * sizeof(double))) return r;
  if (int r = ensure(c, c->d_px, n * sizeof(double))) return r;
  if (int r = ensure(c, c->d_py, n * sizeof(double))) return r;
  if (int r = ensure(c, c->d_pt, n * sizeof(double))) return r;
  if (int r = ensure(c, c->d_hits, n * sizeof(uint32_t))) return r;
  if (int r = ensure(c, c->d_total, R * sizeof(unsigned long long) + 16)) return r;
  if (int r = ensure_pin(c)) return r;
  if (int r = stage_and_upload_runs(c)) return r;
  const bool prof = c->opt_profile == 1, span = c->opt_profile == 2 && c->opt_graph;      // (as run_gmm_full)
  const size_t nprof = c->opt_fused ? 1 : (W > 1 ? W - 1 : 0);
  if (int r = prof_begin(c, nprof > 0 ? nprof : 1)) return r;
  // (the counter reset ahead of the launches and the result copy behind them are plain stream operations: the
  // captured graph holds kernel nodes only, like the GMM path's)
  HIPCHK(c, hipMemsetAsync(c->d_total.p, 0, R * sizeof(unsigned long long), c->stream));
  if (c->opt_graph && !prof) {
    const std::string key = config_key(c, first, count, "mc") + std::to_string(c->num_particles);
    if (!c->graph_mc || key != c->graph_mc_key) {
      if (c->graph_mc) { hipGraphExecDestroy(c->graph_mc); c->graph_mc = nullptr; }
      hipGraph_t g = nullptr;
      HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      int r = enqueue_mc_all(c, first, count, false);
      hipError_t e = hipStreamEndCapture(c->stream, &g);
      if (r) { if (g) hipGraphDestroy(g); return r; }
      HIPCHK(c, e);
      e = hipGraphInstantiate(&c->graph_mc, g, nullptr, nullptr, 0);
      hipGraphDestroy(g);
      HIPCHK(c, e);
      c->graph_mc_key = key;
    }
    if (span) HIPCHK(c, hipEventRecord(c->ev_seq[0], c->stream));
    HIPCHK(c, hipGraphLaunch(c->graph_mc, c->stream));
    if (span) HIPCHK(c, hipEventRecord(c->ev_seq[1], c->stream));
  } else {
    if (int r = enqueue_mc_all(c, first, count, prof)) return r;
  }
  const PinLayout pl = pin_layout(c);
  HIPCHK(c, hipMemcpyAsync((double*)c->h_pin + pl.total, c->d_total.p, R * sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, c->stream));
  prefetch_next_batch(c);          // host chains of the next batch, while the GPU works on this one
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (int r = prof_collect(c, nprof)) return r;
  if (span) {                                          // the graph's span over its W - 1 hot launches (+ the init and count launches: an upper bound)
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev_seq[0], c->ev_seq[1]));
    c->prof_ms = ms; c->prof_launches = (long long)(nprof > 0 ? nprof : 1);
  }
  c->mc_counts.resize(R);
  memcpy(c->mc_counts.data(), (double*)c->h_pin + pl.total, R * sizeof(unsigned long long));
  c->last_mc_count = count;
  c->last_kind = 2;
  return POCS_OK;
}

// ------------------------------------------------------------------------------------------
// text dispatcher
// ------------------------------------------------------------------------------------------
int put(pocs_ctx* c, char* out, size_t cap, const char* text) {
  if (!out || cap == 0) return POCS_OK;
  const size_t n = strlen(text);
  if (n + 1 > cap) { out[0] = 0; return fail(c, POCS_E_BUFFER, "reply needs %zu bytes", n + 1); }
  memcpy(out, text, n + 1);
  return POCS_OK;
}

const char* kHelp =
    "MyCommand        This is an example command\n"
    "ArmaCommand      kept for compatibility (no-op)\n"
    "setAlphas        a1 a2 a3 a4: squared odometry noise coefficients\n"
    "setQ             q: variance of the range sensor noise\n"
    "setNumLandmarks  n\n"
    "setLandmarks     x_0..x_{n-1} y_0..y_{n-1}\n"
    "setNumParticles  n: particles of the MC simulation\n"
    "setInitialCovariance  c00 c01 c02 c10 .. c22 (row major)\n"
    "setPathLength    W\n"
    "setTrajectory    x_0..x_{W-1} y_0..y_{W-1} theta_0..theta_{W-1}\n"
    "setOdometry      r1_0.. tr_0.. r2_0.. (W-1 each)\n"
    "runSimulation    run the MC simulation, replies the collision probability\n"
    "setNumGaussians  k: components of the mixture (1..8)\n"
    "runGMMEstimation sampling-based GMM estimate, replies the collision probability\n"
    "setNumGMMSamples n: samples per waypoint for the GMM estimate\n"
    "setSeed          s: 64-bit seed of the counter-based random streams (new)\n"
    "setFootprint     dx dy half_x half_y (new)\n"
    "addObstacle      cx cy half_x half_y yaw_rad (new)\n"
    "clearObstacles   (new)\n"
    "setBatch         r: independent GMM estimations advanced in lockstep per runGMMEstimation (new)\n"
    "setRunAhead      r: with one run per command, evaluate the next r runs in one launch and serve the following commands from it (new)\n"
    "help             this text\n";

}  // namespace

// ==============================================================================================
// C ABI
// ==============================================================================================
extern "C" {

const char* pocs_version(void) { return POCS_VERSION_STRING; }

int pocs_create(pocs_ctx** out, int device) {
  if (!out) return POCS_E_ARG;
  *out = nullptr;
  pocs_ctx* c = new (std::nothrow) pocs_ctx();
  if (!c) return POCS_E_ARG;
  *out = c;        // returned even on failure so the caller can read pocs_last_error
  memset(&c->sensor, 0, sizeof c->sensor);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(c, POCS_E_DEVICE, "no HIP device available (%s); libpocs has no CPU path",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device < 0 || device >= n) return fail(c, POCS_E_ARG, "device %d out of range (0..%d)", device, n - 1);
  c->device = device;
  HIPCHK(c, hipSetDevice(device));
  HIPCHK(c, hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  for (int g = 0; g < 3; ++g) {
    HIPCHK(c, hipStreamCreateWithFlags(&c->side_stream[g], hipStreamNonBlocking));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[g], hipEventDisableTiming));
  }
  HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIPCHK(c, hipEventCreate(&c->ev_seq[0]));
  HIPCHK(c, hipEventCreate(&c->ev_seq[1]));
  return POCS_OK;
}

#if defined(POCS_TUNING) && defined(POCS_STAMPS)
void pocs_stamps_report();
#endif
void pocs_destroy(pocs_ctx* c) {
  if (!c) return;
#if defined(POCS_TUNING) && defined(POCS_STAMPS)
  if (c->own_stream) { hipSetDevice(c->device); hipStreamSynchronize(c->stream); pocs_stamps_report(); }
#endif
  if (c->own_stream) {
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    drop_graphs(c);
    for (hipEvent_t e : c->events) hipEventDestroy(e);
    for (int g = 0; g < 3; ++g) {
      if (c->side_stream[g]) { hipStreamSynchronize(c->side_stream[g]); hipStreamDestroy(c->side_stream[g]); }
      if (c->ev_join[g]) hipEventDestroy(c->ev_join[g]);
    }
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_seq[0]) hipEventDestroy(c->ev_seq[0]);
    if (c->ev_seq[1]) hipEventDestroy(c->ev_seq[1]);
    DevBuf* all[] = {&c->d_env, &c->d_sensor, &c->d_hdr, &c->d_chain, &c->d_state, &c->d_param,
                     &c->d_moments, &c->d_partial, &c->d_sx, &c->d_sy, &c->d_st, &c->d_flags,
                     &c->d_px, &c->d_py, &c->d_pt, &c->d_hits, &c->d_total, &c->d_ticket, &c->d_tables};
    for (DevBuf* b : all) if (b->p) hipFree(b->p);
    if (c->h_pin) hipHostFree(c->h_pin);
    if (c->h_copy) hipHostFree(c->h_copy);
    for (int q = 0; q < POCS_XCHG_MAX_WORLD; ++q)
      if (c->xchg_peer[q] && c->xchg_peer[q] != c->xchg_own) (void)hipIpcCloseMemHandle(c->xchg_peer[q]);
    if (c->xchg_own) (void)hipFree(c->xchg_own);
    hipStreamDestroy(c->own_stream);
  }
  delete c;
}

const char* pocs_last_error(const pocs_ctx* c) { return c ? c->err.c_str() : "null context"; }

int pocs_set_footprint(pocs_ctx* c, double dx, double dy, double hx, double hy) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (!(hx > 0) || !(hy > 0)) return fail(c, POCS_E_ARG, "footprint half extents must be > 0");
  c->fp.dx = dx; c->fp.dy = dy; c->fp.hx = hx; c->fp.hy = hy;
  c->env_dirty = true;
  return POCS_OK;
}

int pocs_set_obstacles(pocs_ctx* c, const double* boxes, int M) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (M < 0 || M > POCS_MAX_OBSTACLES || (M > 0 && !boxes))
    return fail(c, POCS_E_ARG, "obstacle count %d outside 0..%d", M, POCS_MAX_OBSTACLES);
  for (int m = 0; m < M; ++m)
    if (!(boxes[5 * m + 2] > 0) || !(boxes[5 * m + 3] > 0))
      return fail(c, POCS_E_ARG, "obstacle %d: half extents must be > 0", m);
  c->boxes.assign(boxes, boxes + (size_t)M * 5);
  c->have_obstacles = true;
  c->env_dirty = true;
  return POCS_OK;
}

int pocs_set_alphas(pocs_ctx* c, const double* a, int n) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (n < 1 || n > 4 || !a) return fail(c, POCS_E_ARG, "setAlphas takes 1..4 values (got %d)", n);
  for (int i = 0; i < n; ++i) c->alphas[i] = a[i];
  c->have_alphas = true;
  return POCS_OK;
}

int pocs_set_q(pocs_ctx* c, double q) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (!(q >= 0)) return fail(c, POCS_E_ARG, "Q must be >= 0");
  c->sensor.Q = q; c->have_q = true; c->sensor_dirty = true;
  return POCS_OK;
}

int pocs_set_num_landmarks(pocs_ctx* c, int n) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (n < 0 || n > POCS_MAX_LANDMARKS) return fail(c, POCS_E_ARG, "numLandmarks %d outside 0..%d", n, POCS_MAX_LANDMARKS);
  c->num_landmarks = n; c->have_landmarks = false;
  return POCS_OK;
}

int pocs_set_landmarks(pocs_ctx* c, const double* xy, int n) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (c->num_landmarks < 0) return fail(c, POCS_E_ORDER, "setLandmarks before setNumLandmarks");
  if (n != c->num_landmarks || (n > 0 && !xy)) return fail(c, POCS_E_ARG, "setLandmarks needs 2*%d values", c->num_landmarks);
  c->sensor.L = n;
  for (int i = 0; i < n; ++i) { c->sensor.lx[i] = xy[i]; c->sensor.ly[i] = xy[n + i]; }
  c->have_landmarks = true; c->sensor_dirty = true;
  return POCS_OK;
}

int pocs_set_num_particles(pocs_ctx* c, long long n) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (n < 1) return fail(c, POCS_E_ARG, "numParticles must be >= 1");
  c->num_particles = n;
  return POCS_OK;
}

int pocs_set_initial_covariance(pocs_ctx* c, const double* m9) {
  if (c) touch(c);
  if (!c || !m9) return POCS_E_ARG;
  memcpy(c->cov0, m9, 9 * sizeof(double));
  c->have_cov0 = true;
  return POCS_OK;
}

int pocs_set_path_length(pocs_ctx* c, int W) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (W < 1) return fail(c, POCS_E_ARG, "pathLength must be >= 1");
  if (W != c->W) { c->have_traj = false; c->have_odom = false; }
  c->W = W;
  return POCS_OK;
}

int pocs_set_trajectory(pocs_ctx* c, const double* v, int W) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (c->W < 1) return fail(c, POCS_E_ORDER, "setTrajectory before setPathLength");
  if (W != c->W || !v) return fail(c, POCS_E_ARG, "setTrajectory needs 3*%d values", c->W);
  c->traj.assign(v, v + (size_t)3 * W);
  c->have_traj = true;
  return POCS_OK;
}

int pocs_set_odometry(pocs_ctx* c, const double* v, int Wm1) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (c->W < 1) return fail(c, POCS_E_ORDER, "setOdometry before setPathLength");
  if (Wm1 != c->W - 1 || (Wm1 > 0 && !v)) return fail(c, POCS_E_ARG, "setOdometry needs 3*%d values", c->W - 1);
  c->odom.assign(v, v + (size_t)3 * Wm1);
  c->have_odom = true;
  return POCS_OK;
}

int pocs_set_num_gaussians(pocs_ctx* c, int K) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (K < 1 || K > POCS_MAX_GAUSSIANS) return fail(c, POCS_E_ARG, "numGaussians %d outside 1..%d", K, POCS_MAX_GAUSSIANS);
  c->K = K;
  return POCS_OK;
}

int pocs_set_num_gmm_samples(pocs_ctx* c, long long n) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (n < 1) return fail(c, POCS_E_ARG, "numGMMSamples must be >= 1");
  c->num_gmm = n;
  return POCS_OK;
}

int pocs_set_seed(pocs_ctx* c, uint64_t seed) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  c->seed = seed; c->run_index = 0;
  return POCS_OK;
}

int pocs_set_option(pocs_ctx* c, int option, long long value) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  switch (option) {
    case POCS_OPT_STORE_SAMPLES: c->opt_store = value ? 1 : 0; break;
    case POCS_OPT_MC_FUSED: c->opt_fused = value ? 1 : 0; break;
    case POCS_OPT_USE_GRAPH: c->opt_graph = value ? 1 : 0; break;
    case POCS_OPT_PROFILE:
      if (value < 0 || value > 2) return fail(c, POCS_E_ARG, "POCS_OPT_PROFILE takes 0, 1 or 2");
      c->opt_profile = value;
      break;
    case POCS_OPT_PERSISTENT:
      // the queue-driven whole-call kernel (k_gmm_run) of round 2 was retired in round 3: slower than one launch per
      // waypoint at every batch size measured, and not worth a second summation shape
      if (value) return fail(c, POCS_E_ARG, "POCS_OPT_PERSISTENT: the queue-driven kernel has been retired (DESIGN.md section 5)");
      break;
    case POCS_OPT_LONE_CALL: c->opt_lone = value ? 1 : 0; break;
    case POCS_OPT_SUB_BATCHES:
      if (value < 0 || value > 2) return fail(c, POCS_E_ARG, "sub-batches %lld outside 0..2 (0 = by the call's size; three lost, four gained less than two: DESIGN.md section 5)", value);
      c->opt_groups = value;
      break;
    case POCS_OPT_MC_NONTEMPORAL:
      if (value < -1 || value > 1) return fail(c, POCS_E_ARG, "POCS_OPT_MC_NONTEMPORAL takes -1 (by size), 0 or 1");
      c->opt_mc_nt = value;
      break;
    case POCS_OPT_RUN_AHEAD:
      if (value < 0 || value > 256) return fail(c, POCS_E_ARG, "run-ahead %lld outside 0..256", value);
      c->run_ahead = (int)value;                     // 0 = sized per call (ra_depth)
      break;
    default: return fail(c, POCS_E_ARG, "unknown option %d", option);
  }
  return POCS_OK;
}

int pocs_set_batch(pocs_ctx* c, int runs) {
  if (!c) return POCS_E_ARG;
  touch(c);
  if (runs < 1 || runs > 256) return fail(c, POCS_E_ARG, "batch %d outside 1..256", runs);
  if (c->gmm_open) return fail(c, POCS_E_ORDER, "pocs_set_batch inside a begin/end sequence");
  c->batch = runs;
  return POCS_OK;
}

int pocs_get_batch_probabilities(pocs_ctx* c, double* out, int cap) {
  if (!c || !out) return POCS_E_ARG;
  if (c->ra_internal) {                       // the caller asked for one run at a time
    if (cap < 1 || c->batch_probs.empty()) return fail(c, POCS_E_BUFFER, "need 1 double");
    out[0] = c->batch_probs[(size_t)c->view];
    return 1;
  }
  if ((int)c->batch_probs.size() > cap) return fail(c, POCS_E_BUFFER, "need %zu doubles", c->batch_probs.size());
  memcpy(out, c->batch_probs.data(), c->batch_probs.size() * sizeof(double));
  return (int)c->batch_probs.size();
}

int pocs_select_batch_run(pocs_ctx* c, int run) {
  if (!c) return POCS_E_ARG;
  if (c->ra_internal) return fail(c, POCS_E_ORDER, "pocs_select_batch_run: the last launch was a run-ahead batch (one run per command)");
  if (run < 0 || run >= c->batch_R || c->batch_probs.empty()) return fail(c, POCS_E_ARG, "run %d outside the last batch (0..%d)", run, c->batch_R - 1);
  if (c->last_kind == 1) gmm_select_view(c, run);       // per-waypoint probabilities and moments of that run
  c->view = run;
  return POCS_OK;
}

int pocs_set_shard(pocs_ctx* c, long long first, long long count) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  if (first == -1 && count == -1) { c->shard_first = -1; c->shard_count = -1; return POCS_OK; }   // whole range
  if (first < 0 || count < 0) return fail(c, POCS_E_ARG, "negative shard");
  c->shard_first = first; c->shard_count = count;
  return POCS_OK;
}

int pocs_set_stream(pocs_ctx* c, void* s) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  c->stream = s ? (hipStream_t)s : c->own_stream;
  drop_graphs(c);
  return POCS_OK;
}

int pocs_gmm_bind_moments(pocs_ctx* c, void* dptr, long long len) {
  if (c) touch(c);
  if (!c) return POCS_E_ARG;
  c->ext_moments = (double*)dptr; c->ext_moments_len = dptr ? len : 0;
  drop_graphs(c);
  return POCS_OK;
}

// run* with run-ahead: serve the next cached run, or evaluate the next `run_ahead` runs at once.
static bool ra_can_serve(const pocs_ctx* c, int kind) {
  return c->ra_have > 0 && c->ra_kind == kind && c->batch == 1 && c->view + 1 < c->ra_have;
}
// Runs evaluated per launch when run-ahead is on.  0 (automatic): enough runs to keep the chip busy for the
// launch's fixed cost to fade -- 1.6 x 10^7 mixture samples or 8 x 10^6 particles (what stays in the
// Infinity Cache between two waypoint launches) per launch, at least 8, at most 64: the reference's
// own 200 runs of 10^4 samples go 64 at a time, a 10^6-sample estimation 16 at a time.
static int ra_depth(const pocs_ctx* c, int kind) {
  if (c->run_ahead != 0) return c->run_ahead;
  const long long n = kind == 1 ? c->num_gmm : c->num_particles;
  const long long want = (kind == 1 ? 16000000LL : 8000000LL) / (n > 0 ? n : 1);
  return (int)(want < 8 ? 8 : want > 64 ? 64 : want);
}
static bool ra_wanted(const pocs_ctx* c, int kind) {
  return ra_depth(c, kind) > 1 && c->batch == 1 && c->shard_first < 0 && !c->opt_profile && !c->ext_moments && !c->gmm_open;
}
static void mc_fill_probs(pocs_ctx* c) {
  // getCollisionProportion, MCSimulator.h:324-330 (of the particles this context evaluated)
  const double den = (double)(c->last_mc_count > 0 ? c->last_mc_count : 1);
  c->batch_probs.assign(c->mc_counts.size(), 0.0);
  for (size_t r = 0; r < c->mc_counts.size(); ++r) c->batch_probs[r] = (double)c->mc_counts[r] / den;
}

int pocs_run_gmm_estimation(pocs_ctx* c, double* probability) {
  if (!c) return POCS_E_ARG;
  if (!probability) return fail(c, POCS_E_ARG, "null output");
  HIPCHK(c, hipSetDevice(c->device));
  if (ra_can_serve(c, 1)) {
    gmm_select_view(c, c->view + 1);
    *probability = c->batch_probs[(size_t)c->view];
    return POCS_OK;
  }
  ra_drop(c);
  c->ra_internal = false;
  if (!ra_wanted(c, 1)) return run_gmm_full(c, probability);
  const int depth = ra_depth(c, 1);
  c->batch = depth;
  const int rc = run_gmm_full(c, probability);
  c->batch = 1;
  if (rc == POCS_OK) { c->ra_have = depth; c->ra_kind = 1; c->ra_internal = true; }
  return rc;
}

int pocs_run_simulation(pocs_ctx* c, double* probability) {
  if (!c) return POCS_E_ARG;
  if (!probability) return fail(c, POCS_E_ARG, "null output");
  HIPCHK(c, hipSetDevice(c->device));
  if (ra_can_serve(c, 2)) {
    c->view += 1;
    *probability = c->batch_probs[(size_t)c->view];
    return POCS_OK;
  }
  ra_drop(c);
  c->ra_internal = false;
  const bool ra = ra_wanted(c, 2);
  const int depth = ra_depth(c, 2);
  if (ra) c->batch = depth;
  const int rc = run_mc_local(c);
  c->batch = ra ? 1 : c->batch;
  if (rc) return rc;
  mc_fill_probs(c);
  if (ra) { c->ra_have = depth; c->ra_kind = 2; c->ra_internal = true; }
  *probability = c->batch_probs[0];
  return POCS_OK;
}

int pocs_mc_run_local(pocs_ctx* c, unsigned long long* collided) {
  if (!c) return POCS_E_ARG;
  if (!collided) return fail(c, POCS_E_ARG, "null output");
  HIPCHK(c, hipSetDevice(c->device));
  ra_drop(c);
  c->ra_internal = false;
  if (int r = run_mc_local(c)) return r;
  *collided = c->mc_counts[0];
  return POCS_OK;
}

int pocs_mc_get_batch_counts(pocs_ctx* c, unsigned long long* out, int cap) {
  if (!c || !out) return POCS_E_ARG;
  if (c->ra_internal) {                       // the caller asked for one run at a time
    if (cap < 1 || c->mc_counts.empty()) return fail(c, POCS_E_BUFFER, "need 1 counter");
    out[0] = c->mc_counts[(size_t)c->view];
    return 1;
  }
  if ((int)c->mc_counts.size() > cap) return fail(c, POCS_E_BUFFER, "need %zu counters", c->mc_counts.size());
  memcpy(out, c->mc_counts.data(), c->mc_counts.size() * sizeof(unsigned long long));
  return (int)c->mc_counts.size();
}

int pocs_gmm_begin(pocs_ctx* c) {
  if (!c) return POCS_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  ra_drop(c);
  c->ra_internal = false;
  if (int r = gmm_prepare(c)) return r;
  if (int r = gmm_upload_run(c)) return r;
  if (int r = prof_begin(c, (size_t)c->W)) return r;
  if (int r = enqueue_ticket_reset(c)) return r;
  c->xchg_calls += 1;
  c->gmm_open = true;
  c->last_gmm_wp = -1;
  c->last_gmm_adv = -1;
  return POCS_OK;
}

int pocs_gmm_advance_local(pocs_ctx* c, int w) {
  if (!c) return POCS_E_ARG;
  if (!c->gmm_open) return fail(c, POCS_E_ORDER, "pocs_gmm_advance_local before pocs_gmm_begin");
  if (w != c->last_gmm_wp + 1 || w != c->last_gmm_adv + 1 || w >= c->W)
    return fail(c, POCS_E_ORDER, "advance of waypoint %d out of sequence", w);
  if (int r = enqueue_advance(c, w)) return r;        // folds the (reduced) moments of w-1
  c->last_gmm_adv = w;
  return POCS_OK;
}

int pocs_gmm_sample_local(pocs_ctx* c, int w) {
  if (!c) return POCS_E_ARG;
  if (!c->gmm_open) return fail(c, POCS_E_ORDER, "pocs_gmm_sample_local before pocs_gmm_begin");
  if (w != c->last_gmm_wp + 1 || w >= c->W) return fail(c, POCS_E_ORDER, "waypoint %d out of sequence", w);
  if (w != c->last_gmm_adv) return fail(c, POCS_E_ORDER, "waypoint %d sampled before pocs_gmm_advance_local(%d)", w, w);
  long long first, count;
  if (int r = gmm_shard(c, &first, &count)) return r;
  if (int r = enqueue_step(c, first, count, w, false, c->opt_profile == 1 ? w : -1)) return r;
  c->last_gmm_wp = w;
  c->last_gmm_count = count;
  return POCS_OK;
}

int pocs_gmm_step_local(pocs_ctx* c, int w) {
  if (int r = pocs_gmm_advance_local(c, w)) return r;
  return pocs_gmm_sample_local(c, w);
}

void* pocs_gmm_moments_ptr(pocs_ctx* c, int w) {
  if (!c || w < 0 || w >= c->W || c->K < 1) return nullptr;
  double* m = moments_dev(c);
  return m ? m + (size_t)w * c->batch * c->K * POCS_NMOM : nullptr;
}

int pocs_gmm_moments_len(const pocs_ctx* c) { return (c && c->K > 0) ? c->batch * c->K * POCS_NMOM : 0; }

int pocs_xchg_create(pocs_ctx* c, int world, int rank, void* handle64) {
  if (!c || !handle64) return POCS_E_ARG;
  if (world < 1 || world > POCS_XCHG_MAX_WORLD || rank < 0 || rank >= world)
    return fail(c, POCS_E_ARG, "exchange: world %d / rank %d outside 1..%d", world, rank, POCS_XCHG_MAX_WORLD);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "pocs.h promises a 64-byte handle");
  HIPCHK(c, hipSetDevice(c->device));
  if (!c->xchg_own) {
    // FINE-GRAINED device memory: other GPUs write into it and this GPU polls it inside a running kernel.
    // Ordinary (coarse-grained) allocations are only coherent with other devices at kernel boundaries --
    // a flag once cached in an XCD's L2 could be read stale for ever.
    HIPCHK(c, hipExtMallocWithFlags(&c->xchg_own, POCS_XCHG_BYTES, hipDeviceMallocFinegrained));
    HIPCHK(c, hipMemset(c->xchg_own, 0, POCS_XCHG_BYTES));       // epoch 0 = nothing has landed
    HIPCHK(c, hipDeviceSynchronize());
  }
  c->xchg_world = world; c->xchg_rank = rank; c->xchg_connected = false;
  hipIpcMemHandle_t h;
  HIPCHK(c, hipIpcGetMemHandle(&h, c->xchg_own));
  memcpy(handle64, &h, sizeof h);
  return POCS_OK;
}

int pocs_xchg_connect(pocs_ctx* c, const void* handles, int world) {
  if (!c || !handles) return POCS_E_ARG;
  if (!c->xchg_own || world != c->xchg_world) return fail(c, POCS_E_ORDER, "pocs_xchg_connect before pocs_xchg_create (or another world size)");
  HIPCHK(c, hipSetDevice(c->device));
  for (int q = 0; q < world; ++q) {
    if (q == c->xchg_rank) { c->xchg_peer[q] = c->xchg_own; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + 64 * (size_t)q, sizeof h);
    if (c->xchg_peer[q] && c->xchg_peer[q] != c->xchg_own) { (void)hipIpcCloseMemHandle(c->xchg_peer[q]); c->xchg_peer[q] = nullptr; }
    HIPCHK(c, hipIpcOpenMemHandle(&c->xchg_peer[q], h, hipIpcMemLazyEnablePeerAccess));
  }
  c->xchg_connected = true;
  return POCS_OK;
}

int pocs_gmm_exchange_local(pocs_ctx* c, int w) {
  if (!c) return POCS_E_ARG;
  if (!c->gmm_open) return fail(c, POCS_E_ORDER, "pocs_gmm_exchange_local before pocs_gmm_begin");
  if (!c->xchg_connected) return fail(c, POCS_E_ORDER, "pocs_gmm_exchange_local before pocs_xchg_connect");
  if (w != c->last_gmm_wp || w != c->last_gmm_adv) return fail(c, POCS_E_ORDER, "exchange of waypoint %d out of sequence", w);
  if (c->batch > POCS_XCHG_MAX_RUNS) return fail(c, POCS_E_ARG, "exchange: at most %d runs per call", POCS_XCHG_MAX_RUNS);
  pocs_gmm_launch a;
  fill_gmm_launch(c, &a, 0, 0, w);
  pocs_xchg_dev x;
  memset(&x, 0, sizeof x);
  for (int q = 0; q < c->xchg_world; ++q) x.buf[q] = (double*)c->xchg_peer[q];
  x.world = c->xchg_world; x.rank = c->xchg_rank;
  x.epoch = (c->xchg_calls << 20) | (unsigned long long)(w + 1);
  x.parity = (int)((c->xchg_calls * (unsigned long long)c->W + (unsigned long long)w) & 1ull);
  HIPCHK(c, pocs_launch_gmm_exchange(c->K, a, x, c->stream));
  if (w + 1 < c->W) c->last_gmm_adv = w + 1;          // the exchange launch has built the mixture of w + 1
  return POCS_OK;
}

// sample(w) + exchange(w) in ONE launch: the block that closes a run's waypoint exchanges its moments
// (k_gmm_step, exchange_in_tail) and advances the mixture -- what pocs_gmm_sample_local followed by
// pocs_gmm_exchange_local does with two launches, bit for bit.
int pocs_gmm_sample_exchange_local(pocs_ctx* c, int w) {
  if (!c) return POCS_E_ARG;
  if (!c->gmm_open) return fail(c, POCS_E_ORDER, "pocs_gmm_sample_exchange_local before pocs_gmm_begin");
  if (!c->xchg_connected) return fail(c, POCS_E_ORDER, "pocs_gmm_sample_exchange_local before pocs_xchg_connect");
  if (w != c->last_gmm_wp + 1 || w >= c->W) return fail(c, POCS_E_ORDER, "waypoint %d out of sequence", w);
  if (w != c->last_gmm_adv) return fail(c, POCS_E_ORDER, "waypoint %d sampled before its mixture exists", w);
  if (c->batch > POCS_XCHG_MAX_RUNS) return fail(c, POCS_E_ARG, "exchange: at most %d runs per call", POCS_XCHG_MAX_RUNS);
  long long first, count;
  if (int r = gmm_shard(c, &first, &count)) return r;
  pocs_gmm_launch a;
  fill_gmm_launch(c, &a, first, count, w);
  a.advance_in_tail = (w + 1 < c->W) ? 1 : 0;
  a.exchange_in_tail = 1;
  for (int q = 0; q < c->xchg_world; ++q) a.xchg.buf[q] = (double*)c->xchg_peer[q];
  a.xchg.world = c->xchg_world; a.xchg.rank = c->xchg_rank;
  a.xchg.epoch = (c->xchg_calls << 20) | (unsigned long long)(w + 1);
  a.xchg.parity = (int)((c->xchg_calls * (unsigned long long)c->W + (unsigned long long)w) & 1ull);
  const int slot = c->opt_profile == 1 ? w : -1;
  if (slot >= 0) HIPCHK(c, hipEventRecord(c->events[2 * slot], c->stream));
  HIPCHK(c, pocs_launch_gmm_step(c->K, a, c->stream));
  if (slot >= 0) HIPCHK(c, hipEventRecord(c->events[2 * slot + 1], c->stream));
  c->last_gmm_wp = w;
  c->last_gmm_count = count;
  if (w + 1 < c->W) c->last_gmm_adv = w + 1;
  return POCS_OK;
}

int pocs_gmm_end(pocs_ctx* c, double* probability) {
  if (!c) return POCS_E_ARG;
  if (!c->gmm_open) return fail(c, POCS_E_ORDER, "pocs_gmm_end before pocs_gmm_begin");
  if (c->last_gmm_wp != c->W - 1) return fail(c, POCS_E_ORDER, "pocs_gmm_end after %d of %d waypoints", c->last_gmm_wp + 1, c->W);
  if (!probability) return fail(c, POCS_E_ARG, "null output");
  const PinLayout pl = pin_layout(c);
  HIPCHK(c, hipMemcpyAsync((double*)c->h_pin + pl.moments, moments_dev(c),
                           (size_t)c->W * c->batch * c->K * POCS_NMOM * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync((double*)c->h_pin + pl.total + c->batch + 1, (unsigned*)c->d_ticket.p + POCS_SYNC_ABORT,
                           sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  prefetch_next_batch(c);          // host chains of the next batch, while the queued work drains
  HIPCHK(c, hipStreamSynchronize(c->stream));
  {
    unsigned gave_up = 0;
    memcpy(&gave_up, (double*)c->h_pin + pl.total + c->batch + 1, sizeof gave_up);
    if (gave_up) { c->gmm_open = false; return fail(c, POCS_E_DEVICE, "a bounded wait expired on the device (code %u: 4 = a peer's moments never arrived); results discarded", gave_up); }
  }
  if (int r = prof_collect(c, (size_t)c->W)) return r;
  gmm_combine(c, (double*)c->h_pin + pl.moments, probability);
  c->gmm_open = false;
  return POCS_OK;
}

int pocs_get_path_length(const pocs_ctx* c) { return c ? c->W : POCS_E_ARG; }

int pocs_get_waypoint_probabilities(pocs_ctx* c, double* out, int cap) {
  if (!c || !out) return POCS_E_ARG;
  if ((int)c->probs.size() > cap) return fail(c, POCS_E_BUFFER, "need %zu doubles", c->probs.size());
  memcpy(out, c->probs.data(), c->probs.size() * sizeof(double));
  return (int)c->probs.size();
}

int pocs_get_moments(pocs_ctx* c, int w, double* out, int cap) {
  if (!c || !out) return POCS_E_ARG;
  const int n = c->K * POCS_NMOM;
  if (w < 0 || (size_t)(w + 1) * n > c->last_moments.size()) return fail(c, POCS_E_ARG, "no moments for waypoint %d", w);
  if (cap < n) return fail(c, POCS_E_BUFFER, "need %d doubles", n);
  memcpy(out, &c->last_moments[(size_t)w * n], (size_t)n * sizeof(double));
  return n;
}

static int copy_out(pocs_ctx* c, void* dst, const void* src_dev, size_t bytes, size_t elem, size_t dst_stride);

int pocs_get_gmm_state(pocs_ctx* c, int w, double* means3, double* covs9, double* weights, double* alive) {
  if (!c) return POCS_E_ARG;
  if (w < 0 || w > c->last_gmm_wp || !c->d_state.p) return fail(c, POCS_E_ARG, "no mixture for waypoint %d", w);
  HIPCHK(c, hipSetDevice(c->device));
  std::vector<double> s((size_t)c->K * POCS_STATE_STRIDE);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const double* run_state = (double*)c->d_state.p + (size_t)c->view * c->W * s.size();    // [run][W][K*16]
  if (int r = copy_out(c, s.data(), run_state + (size_t)w * s.size(), s.size() * sizeof(double), 1, 0)) return r;
  for (int k = 0; k < c->K; ++k) {
    if (means3) memcpy(means3 + 3 * k, &s[(size_t)k * POCS_STATE_STRIDE], 3 * sizeof(double));
    if (covs9) memcpy(covs9 + 9 * k, &s[(size_t)k * POCS_STATE_STRIDE + 3], 9 * sizeof(double));
    if (weights) weights[k] = s[(size_t)k * POCS_STATE_STRIDE + 12];
    if (alive) alive[k] = s[(size_t)k * POCS_STATE_STRIDE + 13];
  }
  return c->K;
}

int pocs_get_host_chain(pocs_ctx* c, double* applied3, double* noisy3, double* z, double* mu3, double* cov9) {
  if (!c) return POCS_E_ARG;
  const int steps = c->W - 1, L = c->sensor.L;
  if (steps < 0 || c->h_chain.size() < (size_t)(steps > 0 ? steps : 1) * POCS_CHAIN_STRIDE)
    return fail(c, POCS_E_STATE, "no run yet");
  if (c->view != 0) compute_chain(c, seed_of_run(c, c->batch_base + (uint64_t)c->view));   // h_chain holds run 0's
  for (int i = 0; i < steps; ++i) {
    const double* rec = &c->h_chain[(size_t)i * POCS_CHAIN_STRIDE];
    if (applied3) memcpy(applied3 + 3 * i, rec, 3 * sizeof(double));
    if (noisy3) memcpy(noisy3 + 3 * i, rec + 6, 3 * sizeof(double));
    if (z) memcpy(z + (size_t)L * i, rec + POCS_CHAIN_Z, (size_t)L * sizeof(double));
    if (mu3) memcpy(mu3 + 3 * i, &c->h_mu[(size_t)3 * i], 3 * sizeof(double));
    if (cov9) memcpy(cov9 + 9 * i, &c->h_cov[(size_t)9 * i], 9 * sizeof(double));
  }
  return steps;
}

// Device -> caller memory through the context's own pinned staging buffer, a piece at a time (the runtime
// would otherwise pin the caller's pageable pages on the fly for every call).
#define POCS_COPY_CHUNK (4u << 20)
static int copy_out(pocs_ctx* c, void* dst, const void* src_dev, size_t bytes, size_t elem, size_t dst_stride) {
#if defined(POCS_TUNING) && defined(POCS_PAGEABLE_GETTERS)      // diagnostic build: round 2's getters (the runtime pins the caller's pages per call)
  if (dst_stride == 0 || dst_stride == elem) { HIPCHK(c, hipMemcpy(dst, src_dev, bytes, hipMemcpyDeviceToHost)); return POCS_OK; }
#endif
  if (!c->h_copy) HIPCHK(c, hipHostMalloc(&c->h_copy, POCS_COPY_CHUNK, hipHostMallocDefault));
  for (size_t off = 0; off < bytes; off += POCS_COPY_CHUNK) {
    const size_t n = bytes - off < POCS_COPY_CHUNK ? bytes - off : POCS_COPY_CHUNK;
    HIPCHK(c, hipMemcpy(c->h_copy, (const char*)src_dev + off, n, hipMemcpyDeviceToHost));
    if (dst_stride == 0) memcpy((char*)dst + off, c->h_copy, n);
    else                                             // scatter elements of `elem` bytes `dst_stride` bytes apart
      for (size_t i = 0; i < n / elem; ++i) memcpy((char*)dst + (off / elem + i) * dst_stride, (const char*)c->h_copy + i * elem, elem);
  }
  return POCS_OK;
}
static long long copy_soa_as_aos(pocs_ctx* c, const DevBuf& bx, const DevBuf& by, const DevBuf& bt,
                                 size_t first, long long n, double* aos) {
  const DevBuf* src[3] = {&bx, &by, &bt};
  for (int j = 0; j < 3; ++j)
    if (copy_out(c, aos + j, (const double*)src[j]->p + first, (size_t)n * sizeof(double), sizeof(double), 3 * sizeof(double)) != POCS_OK) return -1;
  return n;
}

long long pocs_copy_gmm_samples(pocs_ctx* c, double* aos, int16_t* flags, long long cap) {
  if (!c) return POCS_E_ARG;
  const long long n = c->last_gmm_count;
  if (!c->opt_store || !c->d_sx.p || c->last_gmm_wp < 0) return fail(c, POCS_E_STATE, "no stored samples");
  if (cap < n) return fail(c, POCS_E_BUFFER, "need room for %lld samples", n);
  if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
    return fail(c, POCS_E_DEVICE, "sync failed");
  const size_t off = (size_t)c->view * (size_t)sample_stride_of(n);          // this run's slice
  if (aos && copy_soa_as_aos(c, c->d_sx, c->d_sy, c->d_st, off, n, aos) < 0) return fail(c, POCS_E_DEVICE, "copy failed");
  if (flags && copy_out(c, flags, (const int16_t*)c->d_flags.p + off, (size_t)n * sizeof(int16_t), 1, 0) != POCS_OK)
    return fail(c, POCS_E_DEVICE, "copy failed");
  return n;
}

long long pocs_copy_particles(pocs_ctx* c, double* aos, uint32_t* hits, long long cap) {
  if (!c) return POCS_E_ARG;
  const long long n = c->last_mc_count;
  if (!c->d_px.p || n <= 0) return fail(c, POCS_E_STATE, "no particles");
  if (cap < n) return fail(c, POCS_E_BUFFER, "need room for %lld particles", n);
  if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
    return fail(c, POCS_E_DEVICE, "sync failed");
  const size_t off = (size_t)c->view * (size_t)sample_stride_of(n);          // this run's slice
  if (aos && copy_soa_as_aos(c, c->d_px, c->d_py, c->d_pt, off, n, aos) < 0) return fail(c, POCS_E_DEVICE, "copy failed");
  if (hits && copy_out(c, hits, (const uint32_t*)c->d_hits.p + off, (size_t)n * sizeof(uint32_t), 1, 0) != POCS_OK)
    return fail(c, POCS_E_DEVICE, "copy failed");
  return n;
}

// Measured streaming-copy bandwidth of this GPU (read + written bytes per second, GB/s): a plain
// 16-B-per-lane copy of `bytes` (rounded down to 16), best of 5 timed with hipEvents on the context's
// stream.  The ceiling the streaming kernels are compared with next to the datasheet's 8 TB/s.
int pocs_measure_copy_bandwidth(pocs_ctx* c, long long bytes, double* gbps) {
  if (!c || !gbps) return POCS_E_ARG;
  if (bytes < 1024) return fail(c, POCS_E_ARG, "copy size too small");
  HIPCHK(c, hipSetDevice(c->device));
  bytes &= ~15LL;
  void *a = nullptr, *b = nullptr;
  HIPCHK(c, hipMalloc(&a, (size_t)bytes));
  if (hipMalloc(&b, (size_t)bytes) != hipSuccess) { (void)hipFree(a); return fail(c, POCS_E_DEVICE, "hipMalloc failed"); }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipMemsetAsync(a, 1, (size_t)bytes, c->stream);
  double best = 0.0;
  int rc = POCS_OK;
  for (int i = 0; i < 6 && rc == POCS_OK; ++i) {
    (void)hipEventRecord(e0, c->stream);
    if (pocs_launch_copy(a, b, bytes, c->stream) != hipSuccess) rc = fail(c, POCS_E_DEVICE, "copy launch failed");
    (void)hipEventRecord(e1, c->stream);
    if (hipEventSynchronize(e1) != hipSuccess) rc = fail(c, POCS_E_DEVICE, "copy failed");
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (i > 0 && ms > 0.f) { const double g = 2.0 * (double)bytes / (ms * 1e-3) / 1e9; if (g > best) best = g; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  *gbps = best;
  return rc;
}

// Measured write-only streaming bandwidth (GB/s written): a plain fill of `bytes`, best of 5.
int pocs_measure_fill_bandwidth(pocs_ctx* c, long long bytes, double* gbps) {
  if (!c || !gbps) return POCS_E_ARG;
  if (bytes < 1024) return fail(c, POCS_E_ARG, "fill size too small");
  HIPCHK(c, hipSetDevice(c->device));
  bytes &= ~15LL;
  void* a = nullptr;
  HIPCHK(c, hipMalloc(&a, (size_t)bytes));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  double best = 0.0;
  int rc = POCS_OK;
  for (int i = 0; i < 6 && rc == POCS_OK; ++i) {
    (void)hipEventRecord(e0, c->stream);
    if (pocs_launch_fill(a, bytes, c->stream) != hipSuccess) rc = fail(c, POCS_E_DEVICE, "fill launch failed");
    (void)hipEventRecord(e1, c->stream);
    if (hipEventSynchronize(e1) != hipSuccess) rc = fail(c, POCS_E_DEVICE, "fill failed");
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (i > 0 && ms > 0.f) { const double g = (double)bytes / (ms * 1e-3) / 1e9; if (g > best) best = g; }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a);
  *gbps = best;
  return rc;
}

// Test hook: the device's table-driven sampler functions on chosen inputs (include/pocs.h).
int pocs_probe_device_math(pocs_ctx* c, int n, const uint32_t* radius_words, const uint32_t* angle_words, const double* headings,
                           double* z0, double* z1, double* sn, double* cs, double* radius2) {
  if (!c || n < 1 || n > (1 << 20) || !radius_words || !angle_words || !headings || !z0 || !z1 || !sn || !cs || !radius2) return c ? fail(c, POCS_E_ARG, "pocs_probe_device_math: 1 <= n <= 2^20, no null pointers") : POCS_E_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  if (int r = upload_tables(c)) return r;
  const size_t nw = (size_t)n * sizeof(uint32_t), nd = (size_t)n * sizeof(double);
  char* buf = nullptr;                         // [wr | wa | x | out 5 n]
  HIPCHK(c, hipMalloc((void**)&buf, 2 * nw + 6 * nd + 64));
  uint32_t* d_wr = (uint32_t*)buf;
  uint32_t* d_wa = d_wr + n;
  double* d_x = (double*)(buf + ((2 * nw + 15) & ~(size_t)15));
  double* d_out = d_x + n;
  int rc = POCS_OK;
  if (hipMemcpy(d_wr, radius_words, nw, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_wa, angle_words, nw, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_x, headings, nd, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(c, POCS_E_DEVICE, "probe upload failed");
  if (rc == POCS_OK && pocs_launch_probe_math((const pocs_tables*)c->d_tables.p, n, d_wr, d_wa, d_x, d_out, c->stream) != hipSuccess)
    rc = fail(c, POCS_E_DEVICE, "probe launch failed");
  if (rc == POCS_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, POCS_E_DEVICE, "probe kernel failed");
  double* dst[5] = {z0, z1, sn, cs, radius2};
  for (int j = 0; j < 5 && rc == POCS_OK; ++j)
    if (hipMemcpy(dst[j], d_out + (size_t)j * n, nd, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(c, POCS_E_DEVICE, "probe download failed");
  (void)hipFree(buf);
  return rc;
}

int pocs_get_sequence_time(pocs_ctx* c, double* ms, int* concurrent) {
  if (!c) return POCS_E_ARG;
  if (ms) *ms = c->seq_ms;
  if (concurrent) *concurrent = c->seq_groups;
  return POCS_OK;
}

// Sharded GMM calls through the library's own exchange: how long the closers of the last begin..end sequence waited
// for the other ranks' moments, over its (run, waypoint) pairs: min, median, max in microseconds.
int pocs_get_exchange_wait(pocs_ctx* c, double* min_median_max_us) {
  if (!c || !min_median_max_us) return POCS_E_ARG;
  if (c->W < 1 || c->batch < 1 || !c->d_ticket.p) return fail(c, POCS_E_STATE, "no GMM call yet");
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)c->batch * (size_t)c->W;
  std::vector<unsigned> v(n);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (int r = copy_out(c, v.data(), (unsigned*)c->d_ticket.p + sync_xwait_offset(c), n * sizeof(unsigned), sizeof(unsigned), sizeof(unsigned))) return r;
  std::sort(v.begin(), v.end());
  min_median_max_us[0] = 0.01 * v.front();
  min_median_max_us[1] = 0.01 * ((n & 1) ? v[n / 2] : 0.5 * ((double)v[n / 2 - 1] + (double)v[n / 2]));
  min_median_max_us[2] = 0.01 * v.back();
  return POCS_OK;
}

int pocs_get_kernel_time(pocs_ctx* c, double* total_ms, long long* launches) {
  if (!c) return POCS_E_ARG;
  if (total_ms) *total_ms = c->prof_ms;
  if (launches) *launches = c->prof_launches;
  return POCS_OK;
}

// The text channel: the grammar (names, token counts, order rules) lives in pocs_command.hpp -- host only, fuzzed
// under sanitizers on the CPU -- and this is the dispatch of a parsed line to the typed setters above.
int pocs_send_command(pocs_ctx* c, const char* line, char* out, size_t cap) {
  if (!c || !line) return POCS_E_ARG;
  if (out && cap) out[0] = 0;
  const pocs_cmd::Shape shape = {c->num_landmarks, c->W};
  const pocs_cmd::Parsed p = pocs_cmd::parse(line, shape);
  if (p.err) return fail(c, p.err, "%s", p.msg.c_str());
  const std::vector<double>& v = p.v;
  switch (p.id) {
    case pocs_cmd::kMyCommand: return put(c, out, cap, "output");                   // mcsimplugin.cpp:225-231
    case pocs_cmd::kArmaCommand: return POCS_OK;                                     // :189-223 (Armadillo demo) -> no-op
    case pocs_cmd::kHelp: return put(c, out, cap, kHelp);
    case pocs_cmd::kSetAlphas: return pocs_set_alphas(c, v.data(), (int)v.size());   // :174-187
    case pocs_cmd::kSetQ: return pocs_set_q(c, v[0]);
    case pocs_cmd::kSetNumLandmarks: return pocs_set_num_landmarks(c, (int)p.n);
    case pocs_cmd::kSetLandmarks: return pocs_set_landmarks(c, v.data(), c->num_landmarks);
    case pocs_cmd::kSetNumParticles: return pocs_set_num_particles(c, p.n);
    case pocs_cmd::kSetInitialCovariance: return pocs_set_initial_covariance(c, v.data());
    case pocs_cmd::kSetPathLength: return pocs_set_path_length(c, (int)p.n);
    case pocs_cmd::kSetTrajectory: return pocs_set_trajectory(c, v.data(), c->W);
    case pocs_cmd::kSetOdometry: return pocs_set_odometry(c, v.data(), c->W - 1);
    case pocs_cmd::kSetNumGaussians: return pocs_set_num_gaussians(c, (int)p.n);
    case pocs_cmd::kSetNumGMMSamples: return pocs_set_num_gmm_samples(c, p.n);
    case pocs_cmd::kSetSeed: return pocs_set_seed(c, (uint64_t)p.seed);
    case pocs_cmd::kSetFootprint: return pocs_set_footprint(c, v[0], v[1], v[2], v[3]);
    case pocs_cmd::kAddObstacle: {
      std::vector<double> b = c->boxes;
      b.insert(b.end(), v.begin(), v.end());
      return pocs_set_obstacles(c, b.data(), (int)(b.size() / 5));
    }
    case pocs_cmd::kClearObstacles: return pocs_set_obstacles(c, nullptr, 0);
    case pocs_cmd::kSetBatch: return pocs_set_batch(c, (int)p.n);
    case pocs_cmd::kSetRunAhead: return pocs_set_option(c, POCS_OPT_RUN_AHEAD, p.n);
    case pocs_cmd::kRunSimulation: case pocs_cmd::kRunGMMEstimation: {               // :75-81, :66-72
      double prob = 0.0;
      const int r = (p.id == pocs_cmd::kRunSimulation) ? pocs_run_simulation(c, &prob) : pocs_run_gmm_estimation(c, &prob);
      if (r) return r;
      char buf[64];
      snprintf(buf, sizeof buf, "%.17g", prob);
      return put(c, out, cap, buf);
    }
    case pocs_cmd::kUnknown: break;
  }
  return fail(c, POCS_E_UNKNOWN_COMMAND, "unknown command '%s'", p.name.c_str());
}

}  // extern "C"
