// pocs_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the hot path.
//
//   k_gmm_run       S1+C1+T1  the whole of runGMMEstimation's sample work -- all W waypoints of every run
//                             of a batch -- in ONE launch: the tasks below handed out from a queue, the
//                             per-run dependency (waypoint w+1 needs the truncated mixture of w) carried
//                             by a `ready` word per run instead of a launch boundary.
//   k_gmm_step      S1+C1+T1  one waypoint of truncateGMM (MCSimulator.h:570-642) in ONE launch, for
//                             every run of a batch of independent estimations (blockIdx.y = run); the
//                             per-waypoint form for callers that exchange moments in between (multi-GPU).
//                             A task (both kernels):
//                             head  log/sector tables, obstacle table and this waypoint's sampler
//                                   parameters -> LDS; exact culling of the obstacle table against
//                                   the mixture's bounding box;
//                             body  GM_Model::sampleNPoints (GM_Model.h:83-116) + checkMatrixCollisions
//                                   (:241-253) + the moment sums (:592-611), fused, one PAIR of
//                                   samples per thread-iteration: a sample is born, tested and folded
//                                   into its component's (n, sum x, sum x x^T) in registers; pose
//                                   and flag are streamed out once (24 B + 2 B);
//                             tail  DPP row sums -> LDS -> one write-through partial row per block ->
//                                   the last block to arrive adds the rows in a fixed order and (one
//                                   GPU) advances the mixture to the next waypoint: truncated
//                                   mean/cov, weights (:597-629), per-component EKF predict/update
//                                   (:766-771, :804-812), Cholesky.
//                             The waypoint loop never returns to the host.
//   k_gmm_advance   T1 tail   the same mixture advance as its own launch (waypoint 0; after the
//                             caller's all-reduce when the samples are sharded over GPUs).
//   k_mc_init       P2+P3     initParticles (:287-297) + first checkParticleCollisions (:333-347)
//   k_mc_step       P1+P3     moveParticles (:300-322) + checkParticleCollisions, one waypoint,
//                             particles streamed through HBM (SoA): 24 B in, 24 B out, u32 RMW.
//   k_mc_fused      P1+P3     same arithmetic, whole roll-out in registers (the controls do not
//                             depend on the particles), 0 B per evaluation.
//   k_mc_count      P3        getCollisionProportion (:324-330): |{hits > 0}|.
//
// Bound: these are FP64-VALU / HBM streaming kernels, no contraction => no MFMA.  Mixture
// parameters, the obstacle table and the 12 KB of log/sector tables are staged in LDS once per
// block (all lanes read the same obstacle record => LDS broadcast; the per-lane component and
// table lookups are 16-byte reads).  Reductions are DPP row sums followed by one LDS pass and a
// per-block partial row; partials are combined in a fixed order so results are bitwise
// reproducible run to run (no float atomics).
#include "pocs_kernels.h"

namespace {

// Row sums by DPP: four steps (pairs, quads, half rows, rows) leave every lane of a 16-lane row
// holding its row's sum.  Every lane has a valid source in all four patterns, so `old` is never
// used; the shape is fixed, hence bitwise reproducible run to run.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double row_sum(double v) {
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);   // row_half_mirror
  v += dpp_f64<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ unsigned row_sum_u32(unsigned v) {
  v += (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false);
  v += (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false);
  return v;
}
// Whole-wave sums (MC count kernel): row sums read back through SGPRs, added in row order.
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
  v = row_sum_u32(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 16) +
         (unsigned)__builtin_amdgcn_readlane((int)v, 32) + (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}

// Stage the log / sector tables (3 KB) into LDS.
__device__ __forceinline__ void stage_tables(const pocs_tables* __restrict__ g, pocs_tables* s_tab) {
  const double* src = reinterpret_cast<const double*>(g);
  double* dst = reinterpret_cast<double*>(s_tab);
  for (int j = threadIdx.x; j < (int)(sizeof(pocs_tables) / sizeof(double)); j += blockDim.x) dst[j] = src[j];
}

// The MC kernels only evaluate the footprint heading: the 4 KB sector table is all they need.
__device__ __forceinline__ void stage_sector_table(const pocs_tables* __restrict__ g, pocs_tables* s_tab) {
  const double* src = &g->sc[0][0];
  double* dst = &s_tab->sc[0][0];
  for (int j = threadIdx.x; j < (int)(sizeof(g->sc) / sizeof(double)); j += blockDim.x) dst[j] = src[j];
}

// Stage the collision world into LDS.  s_obs must hold POCS_MAX_OBSTACLES*POCS_OBS_STRIDE doubles.
__device__ __forceinline__ void stage_env(const pocs_env_dev* __restrict__ env, double* s_obs,
                                          pocs_footprint* s_fp, int* s_M) {
  const int M = env->M;
  for (int i = threadIdx.x; i < M * POCS_OBS_STRIDE; i += blockDim.x) s_obs[i] = env->obs[i];
  if (threadIdx.x == 0) { *s_fp = env->fp; *s_M = M; }
}

// ---------------------------------------------------------------------------------------------
// Hand-offs between workgroups inside a launch (partial rows -> last arriver; mixture state and
// sampler parameters -> the tasks of the next waypoint).  cdna_hip_programming.md Guideline 16,
// form R1: every handed-off byte is stored write-through (`sc1`: a relaxed agent-scope atomic
// store), every storing wave drains its stores (s_waitcnt vmcnt(0)), the block meets, ONE lane
// signals with an agent-scope atomic (ticket add / `ready` store).  The consumer polls or draws
// its ticket relaxed, then ONE agent-scope acquire fence (buffer_inv sc1: this CU's L1) + its
// vmcnt(0) + the block barrier, and only then are the bytes loaded -- with L1-bypassing loads on top
// (relaxed agent-scope atomic loads), so no stale line can be served whatever else shares the CU.
// tests/test_handoff_isa.py disassembles libpocs.so and checks that the emitted ISA has these shapes.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_wt(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_wt(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(
      reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// The sample stream of k_gmm_run.  One launch covers every waypoint, and a run's sample slice is
// rewritten at every waypoint by whichever block -- on whichever XCD -- takes the task: with plain or
// non-temporal stores an older line can still sit dirty in ANOTHER XCD's write-back L2 and reach memory
// after the newer one (seen: stale first-waypoint samples in the final buffer).  Write-through stores
// (`sc1`) leave no dirty line behind; each task drains them before it takes its ticket, and the next
// waypoint's tasks of the run start only behind that, so memory sees the waypoints in order.
// SGPR base + 32-bit lane offset, as the compiler addresses the same stores.
typedef double v2d __attribute__((ext_vector_type(2)));
#ifndef POCS_WT_BITS
#define POCS_WT_BITS "sc1"
#endif
__device__ __forceinline__ void store16_wt(const void* base_uniform, unsigned lane_bytes, v2d v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 " POCS_WT_BITS "\n\ts_nop 1" ::"v"(lane_bytes), "v"(v), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ void store4_wt(const void* base_uniform, unsigned lane_bytes, int v) {
  asm volatile("global_store_dword %0, %1, %2 " POCS_WT_BITS ::"v"(lane_bytes), "v"(v), "s"(base_uniform) : "memory");
}
// a 64-bit value the program knows to be wave-uniform, pinned into scalar registers
__device__ __forceinline__ long long uniform64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// consumer side, ONE lane, after its poll matched / its ticket came back: drop this CU's stale lines
__device__ __forceinline__ void acquire_agent() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the invalidate completes before the barrier releases the readers
}

#if defined(POCS_TASK_STAMPS)     // diagnostic build (tools/task_stamps.sh): where a block of k_gmm_run spends its time
#include <stdio.h>
__device__ unsigned long long g_stamps[16];
#define POCS_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); st_[i] += n_ - last_; last_ = n_; } } while (0)
#define POCS_STAMP_ARGS , unsigned long long (&st_)[12], unsigned long long& last_
#define POCS_STAMP_PASS , st_, last_
#else
#define POCS_STAMP(i) do { } while (0)
#define POCS_STAMP_ARGS
#define POCS_STAMP_PASS
#endif

// LDS scratch of the mixture advance (doubles): state[w-1], moments, chain record, sensor, state[w], param[w];
// and of the speculated component counts.
#define POCS_ADV_SCRATCH(K) ((K) * (2 * POCS_STATE_STRIDE + POCS_NMOM + POCS_PARAM_STRIDE) + POCS_CHAIN_STRIDE + \
                             (int)(sizeof(pocs_sensor) / sizeof(double)))
#define POCS_SPEC_SCRATCH(K) ((K) * (POCS_STATE_STRIDE + 2))

// LDS of the GMM kernels (one struct so that the shared pieces below can be handed around).
template <int K, int TB>
struct gmm_smem {
  static constexpr int NC = K * POCS_NMOM;
  alignas(16) pocs_tables tab;                                   // 12 KB log / sector tables, staged once per block
  alignas(16) double obs[POCS_MAX_OBSTACLES * POCS_OBS_STRIDE];  // obstacle table, staged once per block
  // per task, double-buffered: k_gmm_run stages the NEXT task's copy while the current one is sampled
  alignas(16) double keep[2][POCS_MAX_OBSTACLES * POCS_OBS_STRIDE]; // obstacle table culled for the task
  alignas(16) double par[2][K * POCS_PARAM_STRIDE];                 // sampler parameters of (run, waypoint)
  double red[TB / 16][NC];                                       // one row of sums per 16-lane DPP row
  double part[TB];
  double adv[POCS_ADV_SCRATCH(K)];                               // mixture advance: inputs and outputs
  double spec[POCS_SPEC_SCRATCH(K)];
  int nkeep[2];
  int last;
  unsigned next_task;      // k_gmm_run: the task this block runs next ...
  unsigned next_state;     // ... 1 = dequeued only, 2 = parameters + culled table already staged in buffer buf ^ 1
  unsigned go;
};

// Mixture bookkeeping of waypoint `w` (pocs_gmm_advance_component / pocs_gmm_normalise): every input
// (state[w-1], the reduced moments of w-1, the chain record of step w-1, the sensor) is first brought
// to LDS by the whole block in ONE round trip, lanes < K of wave 0 then take one component each
// (truncated mean / covariance, EKF predict + update, Cholesky) while a lane of wave 1 draws the
// component counts as they will come out unless a factorisation fails; lane 0 normalises and the
// wave writes state[w] / param[w] back write-through.
struct adv_ptrs {
  double *l_prev, *l_mom, *l_ch, *l_sen, *l_next, *l_par;
  double *g_state, *g_param;
  const double *g_prev, *g_mom, *g_ch, *g_sen;
  int ss, ps, NC;
};
__device__ __forceinline__ adv_ptrs advance_ptrs(const pocs_gmm_launch& a, int K, int w, int r, double* scratch) {
  adv_ptrs p;
  constexpr int SEN = (int)(sizeof(pocs_sensor) / sizeof(double));
  p.ss = K * POCS_STATE_STRIDE; p.ps = K * POCS_PARAM_STRIDE; p.NC = K * POCS_NMOM;
  p.l_prev = scratch;
  p.l_mom = p.l_prev + p.ss;
  p.l_ch = p.l_mom + p.NC;
  p.l_sen = p.l_ch + POCS_CHAIN_STRIDE;
  p.l_next = p.l_sen + SEN;
  p.l_par = p.l_next + p.ss;
  // run r of the batch: state/param [r][W][..], moments [W][R][..] (one all-reduce per waypoint
  // covers every run), chain [r][W-1][..]
  p.g_state = a.state + (size_t)r * a.W * p.ss;
  p.g_param = a.param + (size_t)r * a.W * p.ps;
  p.g_prev = p.g_state + (size_t)(w > 0 ? w - 1 : 0) * p.ss;
  p.g_mom = a.moments + ((size_t)(w > 0 ? w - 1 : 0) * a.nruns + r) * p.NC;
  p.g_ch = a.chain + ((size_t)r * (a.W > 1 ? a.W - 1 : 1) + (w > 0 ? w - 1 : 0)) * POCS_CHAIN_STRIDE;
  p.g_sen = reinterpret_cast<const double*>(a.sensor);
  return p;
}

// All threads of the block.  mom_in_lds: l_mom already holds the moments of w-1 (the block has just
// reduced them); otherwise they are read from a.moments (own launch: after the caller's all-reduce).
// state[w-1] may have been written by another block of THIS launch: L1-bypassing loads.
__device__ __forceinline__ void advance_stage(const pocs_gmm_launch& a, int K, int w, int r, double* scratch,
                                              bool mom_in_lds, int tid, int nthreads) {
  const adv_ptrs p = advance_ptrs(a, K, w, r, scratch);
  constexpr int SEN = (int)(sizeof(pocs_sensor) / sizeof(double));
  for (int j = tid; j < p.ss; j += nthreads) p.l_prev[j] = load_wt(&p.g_prev[j]);
  if (w > 0 && !mom_in_lds) for (int j = tid; j < p.NC; j += nthreads) p.l_mom[j] = p.g_mom[j];
  for (int j = tid; j < POCS_CHAIN_STRIDE; j += nthreads) p.l_ch[j] = p.g_ch[j];
  for (int j = tid; j < SEN; j += nthreads) p.l_sen[j] = p.g_sen[j];
}

// wave 0, after advance_stage + barrier: one component per lane
__device__ __forceinline__ void advance_components(const pocs_gmm_launch& a, int K, int w, int r, int lane, double* scratch) {
  const adv_ptrs p = advance_ptrs(a, K, w, r, scratch);
  if (lane < K)
    pocs_gmm_advance_component(lane, p.l_prev, (w == 0) ? nullptr : p.l_mom, p.l_ch, p.l_ch + 3, p.l_ch + POCS_CHAIN_Z,
                               reinterpret_cast<const pocs_sensor*>(p.l_sen), p.l_next, p.l_par);
}

// one lane of ANOTHER wave, meanwhile: the component counts of waypoint w on the premise -- checked by
// advance_finish -- that no Cholesky factorisation fails.  spec = K cumulative counts, K alive flags assumed.
__device__ __forceinline__ void speculate_counts(const pocs_gmm_launch& a, int K, int w, int r, double* scratch, double* spec) {
  const adv_ptrs p = advance_ptrs(a, K, w, r, scratch);
  double* st = spec + 2 * K;                                   // a K x STATE_STRIDE image: only [12], [13] matter
  for (int k = 0; k < K; ++k) {
    const double alive_prev = p.l_prev[k * POCS_STATE_STRIDE + 13];
    const double n = p.l_mom[k * POCS_NMOM];
    const bool alive = alive_prev != 0.0 && n >= 2.0;          // pocs_gmm_advance_component / pocs_truncated_moments
    st[k * POCS_STATE_STRIDE + 12] = alive ? n : 0.0;
    st[k * POCS_STATE_STRIDE + 13] = alive ? alive_prev : 0.0;
    spec[K + k] = st[k * POCS_STATE_STRIDE + 13];
  }
  const int last_alive = pocs_normalise_weights(K, 1, st);
  pocs_component_counts(K, st, last_alive, a.hdr[r].seed, (uint32_t)w, (double)a.n_total, spec, 1);
}

// wave 0, after a barrier: weights, component counts (the speculated ones if their premise held),
// write-through stores of state[w] / param[w], drained.
__device__ __forceinline__ void advance_finish(const pocs_gmm_launch& a, int K, int w, int r, int lane, double* scratch,
                                               const double* spec) {
  const adv_ptrs p = advance_ptrs(a, K, w, r, scratch);
  if (lane == 0) {
    bool use_spec = spec != nullptr;
    if (use_spec) for (int k = 0; k < K; ++k) use_spec = use_spec && (p.l_next[k * POCS_STATE_STRIDE + 13] == spec[K + k]);
    if (use_spec) {
      (void)pocs_normalise_weights(K, 1, p.l_next);
      for (int k = 0; k < K; ++k) p.l_par[k * POCS_PARAM_STRIDE + 9] = spec[k];
    } else {
      pocs_gmm_normalise(K, w > 0, p.l_next, p.l_par, a.hdr[r].seed, (uint32_t)w, (double)a.n_total);
    }
  }
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
  for (int j = lane; j < p.ss; j += 64) store_wt(&p.g_state[(size_t)w * p.ss + j], p.l_next[j]);
  for (int j = lane; j < p.ps; j += 64) store_wt(&p.g_param[(size_t)w * p.ps + j], p.l_par[j]);
  drain_stores();
}

// The whole advance to waypoint w by a block of >= 128 threads (every thread calls it).
__device__ __forceinline__ void advance_block(const pocs_gmm_launch& a, int K, int w, int r, double* adv, double* spec,
                                              bool mom_in_lds, int tid, int nthreads) {
  advance_stage(a, K, w, r, adv, mom_in_lds, tid, nthreads);
  __syncthreads();
  if (tid < 64) advance_components(a, K, w, r, tid, adv);
  else if (tid == 64 && w > 0) speculate_counts(a, K, w, r, adv, spec);
  __syncthreads();
  if (tid < 64) advance_finish(a, K, w, r, tid, adv, w > 0 ? spec : nullptr);
}

__global__ __launch_bounds__(128) void k_gmm_advance(pocs_gmm_launch a, int K) {
  __shared__ double s_adv[POCS_ADV_SCRATCH(POCS_MAX_GAUSSIANS)];
  __shared__ double s_spec[POCS_SPEC_SCRATCH(POCS_MAX_GAUSSIANS)];
  advance_block(a, K, a.waypoint, blockIdx.x, s_adv, s_spec, false, threadIdx.x, 128);      // one block per run
}

// A wave leaves component `k`: its 16-lane row sums of (nFree, nColl, 9 sums) are ADDED to the
// block's LDS rows of that component (every wave owns its rows; a wave meets a component once,
// the add only matters for a component it never touched: + 0) and the thread-private sums restart.
template <int TB, int NC>
__device__ __forceinline__ void flush_component(double (*s_red)[NC], int k, double (&acc)[9], unsigned& nfree,
                                                unsigned& ncoll, int tid) {
  const int row = tid >> 4;
  const bool writer = (tid & 15) == 0;
  const unsigned nf = row_sum_u32(nfree);
  const unsigned nc = row_sum_u32(ncoll);
  double* dst = &s_red[row][k * POCS_NMOM];
  if (writer) { dst[0] += (double)nf; dst[1] += (double)nc; }
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    const double v = row_sum(acc[j]);
    if (writer) dst[2 + j] += v;
    acc[j] = 0.0;
  }
  nfree = 0u; ncoll = 0u;
}

// Once per block: the log / sector tables and the obstacle table -> LDS.
template <int K, int TB>
__device__ __forceinline__ void gmm_stage_static(const pocs_gmm_launch& a, gmm_smem<K, TB>& sm) {
  stage_tables(a.tables, &sm.tab);
  for (int j = threadIdx.x; j < a.M * POCS_OBS_STRIDE; j += TB) sm.obs[j] = a.env->obs[j];
}

// Cull the obstacle table against the bounding box of the mixture staged in par[buf] (ONE wave, all 64
// lanes).  A Box-Muller normal is bounded: u >= 2^-32 gives |z| <= sqrt(64 ln 2) < 6.661
// (pocs_normal_pair_w2; 6.67 leaves 0.1 % for the rounding of radius * cos), so every pose the task can
// draw lies within mean_k +- 6.67 (|L00|, |L10|+|L11|) of some component; an obstacle whose inflated
// box (the broad phase of pocs_box_hit) misses that region is rejected by the broad phase for every
// sample, so dropping it here changes no flag.
template <int K, int TB>
__device__ __forceinline__ void gmm_cull(const pocs_gmm_launch& a, gmm_smem<K, TB>& sm, const int buf, const int lane) {
  const pocs_footprint fp = a.fp;
  const int M = a.M;
  double xlo = 1e300, xhi = -1e300, ylo = 1e300, yhi = -1e300;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double* p = &sm.par[buf][k * POCS_PARAM_STRIDE];
    const double ex = 6.67 * fabs(p[3]), ey = 6.67 * (fabs(p[4]) + fabs(p[5]));
    xlo = fmin(xlo, p[0] - ex); xhi = fmax(xhi, p[0] + ex);
    ylo = fmin(ylo, p[1] - ey); yhi = fmax(yhi, p[1] + ey);
  }
  const double pad = sqrt(fp.dx * fp.dx + fp.dy * fp.dy) + 1e-6;   // footprint centre vs base
  xlo -= pad; xhi += pad; ylo -= pad; yhi += pad;
  bool keep = false;
  if (lane < M) {
    const double* o = &sm.obs[lane * POCS_OBS_STRIDE];
    keep = !(o[0] - o[6] > xhi || o[0] + o[6] < xlo || o[1] - o[7] > yhi || o[1] + o[7] < ylo);
  }
  const unsigned long long mask = __ballot(keep);
  if (keep) {
    const int pos = __popcll(mask & ((1ull << lane) - 1ull));
#pragma unroll
    for (int j = 0; j < POCS_OBS_STRIDE; ++j) sm.keep[buf][pos * POCS_OBS_STRIDE + j] = sm.obs[lane * POCS_OBS_STRIDE + j];
  }
  if (lane == 0) sm.nkeep[buf] = __popcll(mask);
}

// The whole block stages task (w, r) into buffer `buf`: sampler parameters (they may have been
// published by another block of this launch: L1-bypassing loads, behind the caller's acquire) and
// the culled obstacle table.  Ends without a barrier: gmm_task's head has one.
template <int K, int TB>
__device__ __forceinline__ void gmm_stage_task(const pocs_gmm_launch& a, gmm_smem<K, TB>& sm, const int buf, const int w, const int r) {
  const int tid = threadIdx.x;
  for (int j = tid; j < K * POCS_PARAM_STRIDE; j += TB)
    sm.par[buf][j] = load_wt(&a.param[((size_t)r * a.W + w) * (K * POCS_PARAM_STRIDE) + j]);
  __syncthreads();
  if (tid < 64) gmm_cull<K, TB>(a, sm, buf, tid);
}

// k_gmm_run, wave 0 of a block: take the block's NEXT task from the queue and, if its parameters are
// already published (w == 0, or ready[r] >= w: the usual case once the pipeline runs), stage them and
// the culled table into buffer `nbuf` -- so the block goes from one body straight into the next.
// Called twice per task.  FIRST at the start of the body, when the wave has no sample stores in flight
// yet: a wave's loads, atomics and stores complete in issue order, and write-through stores are slow
// to complete, so the same three dependent round trips (queue, `ready`, parameters) cost ~2 us each
// here and ~5 us each behind a body's stores.  AGAIN after the wave's share of the body, if the task
// was not published the first time (state 1): only then does the block still have to wait at the top
// of its loop.
template <int K, int TB>
__device__ __forceinline__ void gmm_prefetch_next(const pocs_gmm_launch& a, gmm_smem<K, TB>& sm, const int nbuf, const int lane,
                                                  const bool dequeue) {
  const unsigned per_wp = (unsigned)a.nruns * (unsigned)a.slices;
  const unsigned total = per_wp * (unsigned)a.W;
  unsigned t = 0u;
  if (dequeue) {
    if (lane == 0) {
      t = __hip_atomic_fetch_add(&a.sync[POCS_SYNC_HEAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(&a.sync[POCS_SYNC_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) t = 0xffffffffu;
    }
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
  } else {
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)sm.next_state) == 2u) return;     // staged the first time
    t = (unsigned)__builtin_amdgcn_readfirstlane((int)sm.next_task);
  }
  unsigned state = 1u;
  if (t < total) {
    const int w = __builtin_amdgcn_readfirstlane((int)(t / per_wp));
    const int r = __builtin_amdgcn_readfirstlane((int)((t - (unsigned)w * per_wp) / (unsigned)a.slices));
    bool ready = (w == 0);
    if (!ready) {
      unsigned have = 0u;
      if (lane == 0) have = __hip_atomic_load(&a.sync[POCS_SYNC_READY + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ready = (unsigned)__builtin_amdgcn_readfirstlane((int)have) >= (unsigned)w;
      if (ready) acquire_agent();                                // the poll matched: ONE acquire, then the loads
    }
    if (ready) {
      for (int j = lane; j < K * POCS_PARAM_STRIDE; j += 64)
        sm.par[nbuf][j] = load_wt(&a.param[((size_t)r * a.W + w) * (K * POCS_PARAM_STRIDE) + j]);
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
      gmm_cull<K, TB>(a, sm, nbuf, lane);
      state = 2u;
    }
  }
  if (lane == 0) { sm.next_task = t; sm.next_state = state; }
}

// ---------------------------------------------------------------------------------------------
// ONE TASK = slice `slot` (of `a.slices`) of run r at waypoint w, by one block:
//   head  sampler parameters of (r, w) -> LDS (they may have been published by another block of this
//         launch: L1-bypassing loads behind the caller's acquire); exact culling of the obstacle
//         table against the mixture's bounding box;
//   body  GM_Model::sampleNPoints (GM_Model.h:83-116) + checkMatrixCollisions (MCSimulator.h:241-253)
//         + the moment sums (:592-611), fused, one PAIR of samples per thread-iteration;
//   tail  DPP row sums -> LDS rows -> ONE write-through partial row (r, slot) -> drain -> ticket (r, w).
// Returns true in the block whose ticket was the last of (r, w) (that block has acquired).
// The arithmetic of a task depends on (r, w, slot, a.slices, a.chunks) only -- not on which kernel
// runs it, which block, or when: k_gmm_step and k_gmm_run give bitwise the same partial rows.
// ---------------------------------------------------------------------------------------------
template <int K, bool STORE, bool WT, int TB>
__device__ __forceinline__ bool gmm_task(const pocs_gmm_launch& a, gmm_smem<K, TB>& sm, const int w, const int r, const int slot,
                                         const int buf POCS_STAMP_ARGS) {
  constexpr int NC = K * POCS_NMOM;
  const int tid = threadIdx.x;
  const pocs_footprint fp = a.fp;

  // ---- head: the task's sampler parameters and culled obstacle table are staged in buffer `buf`
  uint64_t seed = a.hdr[r].seed;
  seed = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(seed >> 32)) << 32) |
         (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)seed);          // scalar registers, provably
  // Samples come in component blocks and a thread's sample indices only grow, so a wave works
  // through the components in order: ONE set of sums per thread (the wave's current component),
  // folded into the block's LDS rows when the wave moves on to the next component.
  double acc[9];
  unsigned nfree = 0u, ncoll = 0u;
  int kcur = 0;                                      // wave-uniform
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = 0.0;
  for (int j = tid; j < (TB / 16) * NC; j += TB) (&sm.red[0][0])[j] = 0.0;
  __syncthreads();
  const double* const s_par = sm.par[buf];
  const double* const s_keep = sm.keep[buf];
  const int nkeep = __builtin_amdgcn_readfirstlane(sm.nkeep[buf]);     // a scalar loop bound for the obstacle loop
  POCS_STAMP(1);

  // ---- body: one PAIR of samples (2j, 2j+1) per thread and iteration -- the pair shares two
  // Philox draws = three Box-Muller pairs (pocs_normal3_pair) and its poses leave as 16-byte stores.
  // a.first is even (checked by the host), so local sample 2*lp is global sample first + 2*lp.
  const long long npairs = (a.count + 1) >> 1;
  const uint64_t pair0 = (uint64_t)(a.first >> 1);
  const double first_d = (double)a.first;
  double cumn[K > 1 ? K - 1 : 1];                   // cumulative component counts (wave-uniform)
#pragma unroll
  for (int j = 0; j < K - 1; ++j) cumn[j] = s_par[j * POCS_PARAM_STRIDE + 9];
  // The loop counter is wave-uniform (SGPRs) and the lane adds its tid: the store addresses are a
  // scalar base per iteration plus a constant 16*tid, no per-lane 64-bit address arithmetic.
  double* const xr = a.x + (size_t)r * a.sample_stride;          // this run's slice (sample_stride is even)
  double* const yr = a.y + (size_t)r * a.sample_stride;
  double* const tr = a.th + (size_t)r * a.sample_stride;
  int16_t* const fr = a.flags + (size_t)r * a.sample_stride;
  const long long c_begin = uniform64(((long long)slot * a.chunks) / a.slices);      // (64-bit division runs on the vector unit)
  const long long c_end = uniform64(((long long)(slot + 1) * a.chunks) / a.slices);
  if (WT && tid < 64) gmm_prefetch_next<K, TB>(a, sm, buf ^ 1, tid, true);       // k_gmm_run: the block's next task, see there
#if !defined(POCS_NO_PRIO_ROTATION)
  // The (up to) four waves of a SIMD -- two of this block, two of the co-resident one -- are arbitrated
  // by priority, then AGE: left alone, the oldest wave of a SIMD runs ~1.7 x faster than the youngest for
  // the whole launch, every task ends with its fast waves idle at the barrier and the SIMDs half empty.
  // Rotating the priority with the iteration gives every wave the same share: they reach the barrier
  // together.  slot = which of the block's two waves on this SIMD (waves v and v + TB/256 share one);
  // the second block of a CU is (observed, speed only) the one dispatched 256 blocks later.
  const int prio_slot = (TB >= 512 ? __builtin_amdgcn_readfirstlane((tid >> 6) / (TB / 256)) : 0) +
                        (TB >= 512 ? 2 : 1) * (int)(((blockIdx.x + gridDim.x * blockIdx.y) >> 8) & 3u);
  int prio_it = prio_slot;
#endif
  for (long long base = c_begin * TB; base < c_end * TB; base += TB) {
#if !defined(POCS_NO_PRIO_ROTATION)
    switch (prio_it++ & 3) {                       // s_setprio takes an immediate
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
#endif
    const long long lp = base + tid;
    const bool live = lp < npairs;                 // a lane past the end computes, masked: the row sums below need every lane
    double zz[2][3];
    uint32_t spare[2];
#if defined(POCS_ABLATE_RNG)          // timing-only builds (tools/ablate.sh): outputs are wrong
    for (int h = 0; h < 2; ++h) { zz[h][0] = (double)(lp & 7) * 0.1; zz[h][1] = (double)(lp & 3) * 0.1; zz[h][2] = 0.05; spare[h] = (uint32_t)lp * 2654435761u; }
#elif defined(POCS_ABLATE_BOXMULLER)
    { const pocs_u32x4 A = pocs_draw(seed, pair0 + lp, (uint32_t)w, POCS_STREAM_GMM, 0u), B = pocs_draw(seed, pair0 + lp, (uint32_t)w, POCS_STREAM_GMM, 1u);
      zz[0][0] = (double)A.x * 0x1p-32; zz[0][1] = (double)A.y * 0x1p-32; zz[0][2] = (double)A.z * 0x1p-32; spare[0] = B.z;
      zz[1][0] = (double)A.w * 0x1p-32; zz[1][1] = (double)B.x * 0x1p-32; zz[1][2] = (double)B.y * 0x1p-32; spare[1] = B.w; }
#else
    // The seed is made opaque once per iteration: otherwise the compiler hoists all 20 Philox round
    // keys (seed + r * Weyl constants) out of the loop and pins 20 SGPRs of a register file that is
    // already spilling; recomputing them costs 2 scalar adds per round.
    uint64_t seed_it = seed;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(seed_it));
#endif
    pocs_normal3_pair(seed_it, pair0 + (uint64_t)lp, (uint32_t)w, POCS_STREAM_GMM, &sm.tab, zz[0], zz[1], &spare[0], &spare[1]);
#endif
    const long long i0 = 2 * lp;
    const bool two = live && (i0 + 1) < a.count;  // false only for the last sample of an odd shard
    const double gbase = first_d + (double)i0;     // global index of sample 2*lp (exact: < 2^53)
    double xs[2], ys[2], ts[2];
    bool hits[2];
    int ks[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // component of the sample (GM_Model.h:87-107: counts[k] samples per component, one block
      // after the other): the first component whose cumulative count exceeds the global index
      const double gidx = gbase + (double)h;
      int k = 0;
#pragma unroll
      for (int j = 0; j < K - 1; ++j) k += (cumn[j] <= gidx) ? 1 : 0;
      const double* p = &s_par[k * POCS_PARAM_STRIDE];
      // mvnrnd (glue_mvnrnd_meat.hpp:134-145): chol_lower * z + mean
      const double x = fma(p[3], zz[h][0], p[0]);
      const double y = fma(p[5], zz[h][1], fma(p[4], zz[h][0], p[1]));
      const double t = fma(p[8], zz[h][2], fma(p[7], zz[h][1], fma(p[6], zz[h][0], p[2])));
#if defined(POCS_ABLATE_COLLIDE)
      const bool hit = x > t;
#else
      const bool hit = pocs_pose_collides(x, y, t, &fp, s_keep, nkeep, &sm.tab);
#endif
      xs[h] = x; ys[h] = y; ts[h] = t; hits[h] = hit; ks[h] = k;
    }
#if defined(POCS_ABLATE_MOMENTS)
    acc[0] += xs[0] + ys[0] + ts[0] + xs[1]; nfree += hits[0] ? 0u : 1u; ncoll += (two && ks[1] == 0) ? 1u : 0u;
#else
    // T1 sums: acc += ind * (x, y, t, xx, xy, xt, yy, yt, tt) with ind = 1.0 for a collision-free
    // sample of the component being accumulated, else 0.0 (fma(1, v, acc) == acc + v, fma(0, v, acc)
    // == acc exactly).  A wave sits in ONE component block except where two blocks meet; the
    // components present in the wave are visited in increasing order (scalar loop), the previous
    // component's sums being flushed to the LDS rows first.
    {
      // sample indices grow with the lane: lane 0 holds the wave's first component, lane 63 its last
      const int klo = __builtin_amdgcn_readfirstlane(live ? ks[0] : K);
      const int khi = (__ballot(live) == ~0ull) ? __builtin_amdgcn_readlane(ks[1], 63) : K - 1;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        if (kk < klo || kk > khi) continue;                                   // scalar compares
        if (kk != kcur) {
          flush_component<TB, NC>(sm.red, kcur, acc, nfree, ncoll, tid);
          kcur = kk;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool sel = (h == 0 ? live : two) && ks[h] == kk;
          const double x = xs[h], y = ys[h], t = ts[h];
          nfree += (sel && !hits[h]) ? 1u : 0u;
          ncoll += (sel && hits[h]) ? 1u : 0u;
          const double ind = (sel && !hits[h]) ? 1.0 : 0.0;
          acc[0] = fma(ind, x, acc[0]);
          acc[1] = fma(ind, y, acc[1]);
          acc[2] = fma(ind, t, acc[2]);
          acc[3] = fma(ind, x * x, acc[3]);
          acc[4] = fma(ind, x * y, acc[4]);
          acc[5] = fma(ind, x * t, acc[5]);
          acc[6] = fma(ind, y * y, acc[6]);
          acc[7] = fma(ind, y * t, acc[7]);
          acc[8] = fma(ind, t * t, acc[8]);
        }
      }
    }
#endif
    if (STORE && live) {
      // Both poses of the pair leave together.  For the last sample of an odd shard the second
      // slot is the pair's unused twin: it lands in the padding element of the run's slice
      // (sample_stride >= count + 1 then) and is never read back.  Written once, never re-read by
      // the kernels: non-temporal, so the stream does not displace the tables / partial rows in L2.
      const size_t ub = 2 * (size_t)base;
      const int fl = (hits[0] ? 1 : 0) | ((two && hits[1]) ? 0x10000 : 0);
      if (WT) {                                    // k_gmm_run: write-through, see store16_wt
        store16_wt(xr + ub, 16u * (unsigned)tid, (v2d){xs[0], xs[1]});
        store16_wt(yr + ub, 16u * (unsigned)tid, (v2d){ys[0], ys[1]});
        store16_wt(tr + ub, 16u * (unsigned)tid, (v2d){ts[0], ts[1]});
        store4_wt(fr + ub, 4u * (unsigned)tid, fl);
      } else {
        __builtin_nontemporal_store((v2d){xs[0], xs[1]}, reinterpret_cast<v2d*>(xr + ub) + tid);
        __builtin_nontemporal_store((v2d){ys[0], ys[1]}, reinterpret_cast<v2d*>(yr + ub) + tid);
        __builtin_nontemporal_store((v2d){ts[0], ts[1]}, reinterpret_cast<v2d*>(tr + ub) + tid);
        __builtin_nontemporal_store(fl, reinterpret_cast<int*>(fr + ub) + tid);
      }
    }
  }

  // ---- tail: the last component's sums -> LDS rows; a fixed-order sum over the TB/16 rows -> the
  // task's write-through partial row; every storing wave drains, the block meets, ONE lane takes the
  // ticket of (r, w); the block that draws the last one acquires.
#if !defined(POCS_NO_PRIO_ROTATION)
  __builtin_amdgcn_s_setprio(0);
#endif
  POCS_STAMP(2);
  if (WT && tid < 64) gmm_prefetch_next<K, TB>(a, sm, buf ^ 1, tid, false);      // k_gmm_run: second try, if need be
  flush_component<TB, NC>(sm.red, kcur, acc, nfree, ncoll, tid);
  __syncthreads();
  POCS_STAMP(3);
  if (tid < NC) {
    double v = sm.red[0][tid];
#pragma unroll 8
    for (int q = 1; q < TB / 16; ++q) v += sm.red[q][tid];
    store_wt(&a.partial[((size_t)r * a.slices + slot) * NC + tid], v);
  }
  drain_stores();
  __syncthreads();
  POCS_STAMP(4);
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(&a.ticket[(size_t)r * a.W + w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == (unsigned)a.slices - 1u) ? 1 : 0;
    if (last) acquire_agent();
    sm.last = last;
  }
  __syncthreads();
  POCS_STAMP(5);
  return __builtin_amdgcn_readfirstlane(sm.last) != 0;
}

// The block that drew the last ticket of (r, w): the a.slices partial rows of the run, read back past
// L1 and added in a fixed order (slice q of column c sums rows q, q+S, q+2S, ...; then slices in
// order) -> moments[w][r]; with `advance`, the mixture of waypoint w+1 right away (these ARE the
// global moments on one GPU) -- and, with `publish`, ready[r] = w + 1 for the tasks waiting for it.
template <int K, int TB>
__device__ __forceinline__ void gmm_finish(const pocs_gmm_launch& a, gmm_smem<K, TB>& sm, const int w, const int r,
                                           const bool advance, const bool publish) {
  constexpr int NC = K * POCS_NMOM;
  constexpr int S = TB / NC;
  const int tid = threadIdx.x;
  const int q = tid / NC, c = tid - q * NC;
  double v = 0.0;
  if (q < S) {
    const int nb = a.slices;
    for (int b = q; b < nb; b += 8 * S) {        // 8 loads in flight, added in row order
      double rows[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int bb = b + u * S;
        rows[u] = (bb < nb) ? load_wt(&a.partial[((size_t)r * nb + bb) * NC + c]) : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) v += rows[u];
    }
  }
  sm.part[tid] = v;
  __syncthreads();
  double* const l_mom = advance_ptrs(a, K, w + 1, r, sm.adv).l_mom;
  if (tid < NC) {
    double tot = sm.part[tid];
    for (int sl = 1; sl < S; ++sl) tot += sm.part[sl * NC + tid];
    a.moments[((size_t)w * a.nruns + r) * NC + tid] = tot;
    l_mom[tid] = tot;
  }
  if (advance) {
    advance_block(a, K, w + 1, r, sm.adv, sm.spec, true, tid, TB);       // starts with a barrier after staging
    if (publish && tid == 0)                                              // wave 0 has drained its state / param stores
      __hip_atomic_store(&a.sync[POCS_SYNC_READY + r], (unsigned)(w + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One waypoint as its own launch: grid = (slices, runs), block (j, r) = task (a.waypoint, r, j).  The
// per-waypoint path of a caller that exchanges the moments between waypoints (sharded over GPUs).
template <int K, bool STORE, int TB>
__global__ __launch_bounds__(TB, POCS_GMM_BLOCKS_PER_CU * TB / 256) void k_gmm_step(pocs_gmm_launch a) {
  __shared__ gmm_smem<K, TB> sm;
  gmm_stage_static(a, sm);
  const int w = a.waypoint, r = blockIdx.y;
#if defined(POCS_TASK_STAMPS)
  unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = 0;
#endif
  gmm_stage_task<K, TB>(a, sm, 0, w, r);
  if (gmm_task<K, STORE, false, TB>(a, sm, w, r, (int)blockIdx.x, 0 POCS_STAMP_PASS))     // a launch of its own per waypoint: streaming stores
    gmm_finish<K, TB>(a, sm, w, r, a.advance_in_tail != 0, false);
}

// ---------------------------------------------------------------------------------------------
// The whole run in ONE launch: every task (w, r, j) of the call's W waypoints x R runs x S slices,
// handed out in that order from a queue (one returning atomic per task).  A task of waypoint w > 0
// waits for `ready[r] >= w`, published by the block that closed (r, w-1) -- a task handed out earlier
// to a block that is running, so the wait always ends, however many blocks are resident.  Nothing
// synchronises the grid: while the last arriver of a run reduces and advances its mixture, the other
// blocks are already on tasks of other runs (R * S is ~1.5 x the resident blocks), and the next
// waypoint's tasks of this run find their parameters published when their turn comes.  Every wait is
// bounded (2 s): a block that gives up sets the call's abort word and every block leaves at its next
// dequeue; the host reports POCS_E_DEVICE.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool wait_ready(unsigned* ready, unsigned need, unsigned* abort_word) {
  const unsigned long long t0 = wall_clock64();                         // 100 MHz
  unsigned polls = 0;
  while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(4);
    if ((++polls & 255u) == 0u) {
      if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      if (wall_clock64() - t0 > 200000000ull) {                         // 2 s
        __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
  return true;
}

template <int K, bool STORE, int TB>
__global__ __launch_bounds__(TB, POCS_GMM_BLOCKS_PER_CU * TB / 256) void k_gmm_run(pocs_gmm_launch a) {
  __shared__ gmm_smem<K, TB> sm;
  gmm_stage_static(a, sm);
  const int tid = threadIdx.x;
  const unsigned per_wp = (unsigned)a.nruns * (unsigned)a.slices;
  const unsigned total = per_wp * (unsigned)a.W;
#if defined(POCS_TASK_STAMPS)
  unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = wall_clock64();
  const unsigned long long t_begin_ = last_;
#endif
  if (tid == 0) {                                                       // the block's first task
    unsigned t = __hip_atomic_fetch_add(&a.sync[POCS_SYNC_HEAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (__hip_atomic_load(&a.sync[POCS_SYNC_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) t = 0xffffffffu;
    sm.next_task = t;
    sm.next_state = 1u;
  }
  __syncthreads();
  int buf = 0;
  for (;;) {
    const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)sm.next_task);     // block-uniform, and provably so
    const unsigned state = (unsigned)__builtin_amdgcn_readfirstlane((int)sm.next_state);
    if (t >= total) break;
    // (integer division runs on the vector unit: pin the quotients back into scalar registers)
    const int w = __builtin_amdgcn_readfirstlane((int)(t / per_wp));
    const unsigned rem = t - (unsigned)w * per_wp;
    const int r = __builtin_amdgcn_readfirstlane((int)(rem / (unsigned)a.slices));
    const int slot = (int)rem - r * a.slices;
#if defined(POCS_TASK_STAMPS)
    if (tid == 0 && state == 2u) st_[10] += 1;
    POCS_STAMP(7);                                                      // loop-end barrier + decode
#endif
    if (state != 2u) {                                                  // not staged by the prefetch: wait, then stage
      if (w > 0) {
        if (tid == 0) {
          const unsigned go = wait_ready(&a.sync[POCS_SYNC_READY + r], (unsigned)w, &a.sync[POCS_SYNC_ABORT]) ? 1u : 0u;
          if (go) acquire_agent();
          sm.go = go;
        }
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane((int)sm.go) == 0) break;
      }
#if defined(POCS_TASK_STAMPS)
      POCS_STAMP(11);                                                   // wait for `ready` + acquire
#endif
      gmm_stage_task<K, TB>(a, sm, buf, w, r);
    }
    POCS_STAMP(0);
    if (gmm_task<K, STORE, true, TB>(a, sm, w, r, slot, buf POCS_STAMP_PASS)) {
      gmm_finish<K, TB>(a, sm, w, r, w + 1 < a.W, true);
      POCS_STAMP(6);
#if defined(POCS_TASK_STAMPS)
      if (tid == 0) st_[8] += 1;
#endif
    }
    __syncthreads();
#if defined(POCS_TASK_STAMPS)
    if (tid == 0) st_[9] += 1;
#endif
    buf ^= 1;
  }
#if defined(POCS_TASK_STAMPS)
  if (tid == 0) {
    const unsigned long long life_ = wall_clock64() - t_begin_;
    for (int i = 0; i < 12; ++i) atomicAdd(&g_stamps[i], st_[i]);
    atomicAdd(&g_stamps[12], life_);
    atomicAdd(&g_stamps[13], 1ull);
  }
#endif
}

// MC kernels: blockIdx.y = run of the batch (its own seed, its own noisy controls, its own slice
// of the particle arrays).
struct mc_run_view {
  uint64_t seed;
  const double* chain;
  double* x; double* y; double* th;
  uint32_t* hits;
};
__device__ __forceinline__ mc_run_view mc_view(const pocs_mc_launch& a) {
  const int r = blockIdx.y;
  const size_t o = (size_t)r * (size_t)a.stride;
  mc_run_view v;
  v.seed = a.hdr[r].seed;
  v.chain = a.chain + (size_t)r * (a.W > 1 ? a.W - 1 : 1) * POCS_CHAIN_STRIDE;
  v.x = a.x + o; v.y = a.y + o; v.th = a.th + o; v.hits = a.hits + o;
  return v;
}

__global__ __launch_bounds__(POCS_BLOCK) void k_mc_init(pocs_mc_launch a) {
  __shared__ double s_obs[POCS_MAX_OBSTACLES * POCS_OBS_STRIDE];
  __shared__ pocs_footprint s_fp;
  __shared__ int s_M;
  __shared__ pocs_tables s_tab;
  stage_env(a.env, s_obs, &s_fp, &s_M);
  stage_sector_table(a.tables, &s_tab);
  __syncthreads();
  const mc_run_view v = mc_view(a);
  const pocs_footprint fp = s_fp;
  const int M = s_M;
  const long long stride = (long long)gridDim.x * POCS_BLOCK;
  for (long long i = (long long)blockIdx.x * POCS_BLOCK + threadIdx.x; i < a.count; i += stride) {
    double z[3];
    uint32_t spare;
    pocs_normal3(v.seed, (uint64_t)(a.first + i), 0u, POCS_STREAM_MCINIT, z, &spare);
    const double x = fma(a.L0[0], z[0], a.mu0[0]);
    const double y = fma(a.L0[2], z[1], fma(a.L0[1], z[0], a.mu0[1]));
    const double t = fma(a.L0[5], z[2], fma(a.L0[4], z[1], fma(a.L0[3], z[0], a.mu0[2])));
    v.x[i] = x; v.y[i] = y; v.th[i] = t;
    v.hits[i] = pocs_pose_collides(x, y, t, &fp, s_obs, M, &s_tab) ? 1u : 0u;
  }
}

// NT: non-temporal accesses, chosen by the host when the particle state of the batch does not fit
// the 256 MB Infinity Cache anyway (the stream then runs faster past the caches; when it does fit,
// plain accesses keep it there between waypoint launches).  One particle per thread and iteration:
// a two-particle version with 16-byte accesses measured 12 % slower in cache, 7 % faster out of it.
template <bool NT>
__global__ __launch_bounds__(POCS_BLOCK) void k_mc_step(pocs_mc_launch a) {
  __shared__ double s_obs[POCS_MAX_OBSTACLES * POCS_OBS_STRIDE];
  __shared__ pocs_footprint s_fp;
  __shared__ int s_M;
  __shared__ pocs_tables s_tab;
  stage_env(a.env, s_obs, &s_fp, &s_M);
  stage_sector_table(a.tables, &s_tab);
  __syncthreads();
  const mc_run_view v = mc_view(a);
  const pocs_footprint fp = s_fp;
  const int M = s_M;
  const double* u = v.chain + (size_t)a.step * POCS_CHAIN_STRIDE + 6;
  const double u0 = u[0], u1 = u[1], u2 = u[2];
  const long long stride = (long long)gridDim.x * POCS_BLOCK;
  for (long long i = (long long)blockIdx.x * POCS_BLOCK + threadIdx.x; i < a.count; i += stride) {
    const double x = NT ? __builtin_nontemporal_load(v.x + i) : v.x[i];
    const double y = NT ? __builtin_nontemporal_load(v.y + i) : v.y[i];
    const double t = NT ? __builtin_nontemporal_load(v.th + i) : v.th[i];
    double sn, cs;
    pocs_sincos(t + u0, &sn, &cs);
    const double nx = fma(u1, cs, x);
    const double ny = fma(u1, sn, y);
    const double nt = pocs_wrap_angle(t + u0 + u2);
    if (NT) {
      __builtin_nontemporal_store(nx, v.x + i); __builtin_nontemporal_store(ny, v.y + i); __builtin_nontemporal_store(nt, v.th + i);
    } else {
      v.x[i] = nx; v.y[i] = ny; v.th[i] = nt;
    }
    if (pocs_pose_collides(nx, ny, nt, &fp, s_obs, M, &s_tab)) v.hits[i] += 1u;
  }
}

__global__ __launch_bounds__(POCS_BLOCK) void k_mc_fused(pocs_mc_launch a) {
  __shared__ double s_obs[POCS_MAX_OBSTACLES * POCS_OBS_STRIDE];
  __shared__ pocs_footprint s_fp;
  __shared__ int s_M;
  __shared__ pocs_tables s_tab;
  stage_env(a.env, s_obs, &s_fp, &s_M);
  stage_sector_table(a.tables, &s_tab);
  __syncthreads();
  const mc_run_view v = mc_view(a);
  const pocs_footprint fp = s_fp;
  const int M = s_M;
  const long long stride = (long long)gridDim.x * POCS_BLOCK;
  for (long long i = (long long)blockIdx.x * POCS_BLOCK + threadIdx.x; i < a.count; i += stride) {
    double z[3];
    uint32_t spare;
    pocs_normal3(v.seed, (uint64_t)(a.first + i), 0u, POCS_STREAM_MCINIT, z, &spare);
    double x = fma(a.L0[0], z[0], a.mu0[0]);
    double y = fma(a.L0[2], z[1], fma(a.L0[1], z[0], a.mu0[1]));
    double t = fma(a.L0[5], z[2], fma(a.L0[4], z[1], fma(a.L0[3], z[0], a.mu0[2])));
    unsigned h = pocs_pose_collides(x, y, t, &fp, s_obs, M, &s_tab) ? 1u : 0u;
    for (int s = 0; s < a.step; ++s) {
      const double* u = v.chain + (size_t)s * POCS_CHAIN_STRIDE + 6;   // wave-uniform
      const double u0 = u[0], u1 = u[1], u2 = u[2];
      double sn, cs;
      pocs_sincos(t + u0, &sn, &cs);
      x = fma(u1, cs, x);
      y = fma(u1, sn, y);
      t = pocs_wrap_angle(t + u0 + u2);
      h += pocs_pose_collides(x, y, t, &fp, s_obs, M, &s_tab) ? 1u : 0u;
    }
    v.x[i] = x; v.y[i] = y; v.th[i] = t;
    v.hits[i] = h;
  }
}

__global__ __launch_bounds__(POCS_BLOCK) void k_mc_count(pocs_mc_launch a) {
  __shared__ unsigned s_w[POCS_BLOCK / 64];
  const mc_run_view v = mc_view(a);
  unsigned c = 0;
  const long long stride = (long long)gridDim.x * POCS_BLOCK;
  for (long long i = (long long)blockIdx.x * POCS_BLOCK + threadIdx.x; i < a.count; i += stride)
    c += v.hits[i] > 0u ? 1u : 0u;
  c = wave_sum_u32(c);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < POCS_BLOCK / 64; ++w) t += s_w[w];
    if (t) atomicAdd(&a.total[blockIdx.y], t);      // integer atomic: order independent, exact
  }
}

template <int K>
hipError_t launch_gmm_k(const pocs_gmm_launch& a, hipStream_t s) {
  constexpr int TB = POCS_GMM_BLOCK_OF(K);
  if (a.store) hipLaunchKernelGGL((k_gmm_step<K, true, TB>), dim3(a.slices, a.nruns), dim3(TB), 0, s, a);
  else         hipLaunchKernelGGL((k_gmm_step<K, false, TB>), dim3(a.slices, a.nruns), dim3(TB), 0, s, a);
  return hipGetLastError();
}
template <int K>
hipError_t launch_gmm_run_k(int nblk, const pocs_gmm_launch& a, hipStream_t s) {
  constexpr int TB = POCS_GMM_BLOCK_OF(K);
  if (a.store) hipLaunchKernelGGL((k_gmm_run<K, true, TB>), dim3(nblk), dim3(TB), 0, s, a);
  else         hipLaunchKernelGGL((k_gmm_run<K, false, TB>), dim3(nblk), dim3(TB), 0, s, a);
  return hipGetLastError();
}

}  // namespace

hipError_t pocs_launch_gmm_step(int K, const pocs_gmm_launch& a, hipStream_t s) {
  switch (K) {
    case 1: return launch_gmm_k<1>(a, s);
    case 2: return launch_gmm_k<2>(a, s);
    case 3: return launch_gmm_k<3>(a, s);
    case 4: return launch_gmm_k<4>(a, s);
    case 5: return launch_gmm_k<5>(a, s);
    case 6: return launch_gmm_k<6>(a, s);
    case 7: return launch_gmm_k<7>(a, s);
    case 8: return launch_gmm_k<8>(a, s);
    default: return hipErrorInvalidValue;
  }
}

#if defined(POCS_TASK_STAMPS)
static hipError_t launch_gmm_run_plain(int K, int nblk, const pocs_gmm_launch& a, hipStream_t s);
hipError_t pocs_launch_gmm_run(int K, int nblk, const pocs_gmm_launch& a, hipStream_t s) {
  unsigned long long z[16] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z);
  const hipError_t e = launch_gmm_run_plain(K, nblk, a, s);
  (void)hipStreamSynchronize(s);
  unsigned long long h[16];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof h) == hipSuccess && h[13] > 0) {
    static const char* names[7] = {"stage", "head", "body", "flush", "partial+drain", "ticket", "finish"};
    const double nb = (double)h[13], tasks = (double)h[9];
    fprintf(stderr, "stamps: %g blocks, %g tasks (%g finishes, %.0f %% prefetched), mean block lifetime %.1f us; per task (us): loop-end %.2f wait+acquire %.2f",
            nb, tasks, (double)h[8], 100.0 * (double)h[10] / tasks, 0.01 * h[12] / nb, 0.01 * (double)h[7] / tasks, 0.01 * (double)h[11] / tasks);
    for (int i = 0; i < 6; ++i) fprintf(stderr, " %s %.2f", names[i], 0.01 * (double)h[i] / tasks);
    fprintf(stderr, "; per finish: %.2f us\n", h[8] ? 0.01 * (double)h[6] / (double)h[8] : 0.0);
  }
  return e;
}
static hipError_t launch_gmm_run_plain(int K, int nblk, const pocs_gmm_launch& a, hipStream_t s) {
#else
hipError_t pocs_launch_gmm_run(int K, int nblk, const pocs_gmm_launch& a, hipStream_t s) {
#endif
  switch (K) {
    case 1: return launch_gmm_run_k<1>(nblk, a, s);
    case 2: return launch_gmm_run_k<2>(nblk, a, s);
    case 3: return launch_gmm_run_k<3>(nblk, a, s);
    case 4: return launch_gmm_run_k<4>(nblk, a, s);
    case 5: return launch_gmm_run_k<5>(nblk, a, s);
    case 6: return launch_gmm_run_k<6>(nblk, a, s);
    case 7: return launch_gmm_run_k<7>(nblk, a, s);
    case 8: return launch_gmm_run_k<8>(nblk, a, s);
    default: return hipErrorInvalidValue;
  }
}

// Plain streaming copy: the measured HBM ceiling the streaming kernels are compared with next to
// the datasheet peak (bench.py "copy_GBps").  Four 16-byte non-temporal loads in flight per lane,
// 8192 blocks: the best of the variants in tools/ubench/copy_rates.hip (6.0-6.2 TB/s read + written;
// one plain load per lane on 2048 blocks stops at 4.9).
__global__ __launch_bounds__(POCS_BLOCK) void k_copy(const double2* __restrict__ src, double2* __restrict__ dst, long long n) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const v2d* s = reinterpret_cast<const v2d*>(src);
  v2d* d = reinterpret_cast<v2d*>(dst);
  const long long stride = (long long)gridDim.x * POCS_BLOCK * 4;
  for (long long i = (long long)blockIdx.x * POCS_BLOCK * 4 + threadIdx.x; i < n; i += stride) {
    v2d v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * POCS_BLOCK < n) v[u] = __builtin_nontemporal_load(s + i + u * POCS_BLOCK);
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * POCS_BLOCK < n) __builtin_nontemporal_store(v[u], d + i + u * POCS_BLOCK);
  }
}
hipError_t pocs_launch_copy(const void* src, void* dst, long long bytes, hipStream_t s) {
  hipLaunchKernelGGL(k_copy, dim3(8192), dim3(POCS_BLOCK), 0, s, (const double2*)src, (double2*)dst, bytes / 16);
  return hipGetLastError();
}

hipError_t pocs_launch_gmm_advance(int K, const pocs_gmm_launch& a, hipStream_t s) {
  hipLaunchKernelGGL(k_gmm_advance, dim3(a.nruns), dim3(128), 0, s, a, K);
  return hipGetLastError();
}
hipError_t pocs_launch_mc_init(int nblk, const pocs_mc_launch& a, hipStream_t s) {
  hipLaunchKernelGGL(k_mc_init, dim3(nblk, a.nruns), dim3(POCS_BLOCK), 0, s, a);
  return hipGetLastError();
}
hipError_t pocs_launch_mc_step(int nblk, const pocs_mc_launch& a, hipStream_t s) {
  if (a.nontemporal) hipLaunchKernelGGL(k_mc_step<true>, dim3(nblk, a.nruns), dim3(POCS_BLOCK), 0, s, a);
  else               hipLaunchKernelGGL(k_mc_step<false>, dim3(nblk, a.nruns), dim3(POCS_BLOCK), 0, s, a);
  return hipGetLastError();
}
hipError_t pocs_launch_mc_fused(int nblk, const pocs_mc_launch& a, hipStream_t s) {
  hipLaunchKernelGGL(k_mc_fused, dim3(nblk, a.nruns), dim3(POCS_BLOCK), 0, s, a);
  return hipGetLastError();
}
hipError_t pocs_launch_mc_count(int nblk, const pocs_mc_launch& a, hipStream_t s) {
  hipLaunchKernelGGL(k_mc_count, dim3(nblk, a.nruns), dim3(POCS_BLOCK), 0, s, a);
  return hipGetLastError();
}
