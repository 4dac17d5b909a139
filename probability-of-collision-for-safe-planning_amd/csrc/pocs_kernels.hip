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
//                                   the mixture's bounding box, the kept records' broad phase
//                                   tightened to the task's range of headings;
//                             body  GM_Model::sampleNPoints (GM_Model.h:83-116) + checkMatrixCollisions
//                                   (:241-253) + the moment sums (:592-611), fused, one PAIR of
//                                   samples per thread-iteration: a sample is born, tested and folded
//                                   into its component's (n, sum x, sum x x^T) in registers; pose
//                                   and flag are streamed out once (24 B + 2 B).  A wave whose 128
//                                   samples lie in one component block (nearly always) runs the
//                                   iteration's scalar-component form in an inner loop of its own;
//                             tail  DPP row sums -> LDS -> one write-through partial row per block ->
//                                   the last block to arrive adds the rows in a fixed order and
//                                   advances the mixture to the next waypoint: truncated mean/cov,
//                                   weights (:597-629), per-component EKF predict/update (:766-771,
//                                   :804-812), Cholesky -- on one GPU right away, sharded after it has
//                                   exchanged the run's moments with the other ranks (IPC slots, one
//                                   hop over xGMI) in the same tail.
//                             The waypoint loop never returns to the host.
//   k_gmm_advance   T1 tail   the same mixture advance as its own launch (waypoint 0; after the
//                             caller's all-reduce when the shards exchange their moments that way).
//   k_gmm_exchange  T1 tail   exchange + advance as their own launch (the step API's two-launch form).
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
#include <type_traits>

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

// Stage the log / sector tables (12 KB) into LDS.
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
// the same addressing for the per-waypoint kernel's non-temporal stream (the compiler, left to itself,
// builds a 64-bit address per lane and store: four vector adds per iteration)
__device__ __forceinline__ void store16_nt(const void* base_uniform, unsigned lane_bytes, v2d v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_bytes), "v"(v), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ void store4_nt(const void* base_uniform, unsigned lane_bytes, int v) {
  asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(lane_bytes), "v"(v), "s"(base_uniform) : "memory");
}
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

#if defined(POCS_TASK_STAMPS)     // diagnostic build (tools/task_stamps.sh): where a WAVE of k_gmm_run spends its time
#include <stdio.h>
__device__ unsigned long long g_stamps[32];
#define POCS_STAMP(i) do { const unsigned long long n_ = wall_clock64(); st_[i] += n_ - last_; last_ = n_; } while (0)
#else
#define POCS_STAMP(i) do { } while (0)
#endif

// LDS scratch of the mixture advance (doubles): state[w-1], moments, chain record, sensor, state[w], param[w];
// and of the speculated component counts.
#define POCS_ADV_SCRATCH(K) ((K) * (2 * POCS_STATE_STRIDE + POCS_NMOM + POCS_PARAM_STRIDE) + POCS_CHAIN_STRIDE + \
                             (int)(sizeof(pocs_sensor) / sizeof(double)))
#define POCS_SPEC_SCRATCH(K) ((K) * (POCS_STATE_STRIDE + 2))

// LDS of the GMM kernels.  NB = task buffers: 1 for k_gmm_step (one task per block), 3 for k_gmm_run
// (the waves of a block drift up to a task apart, see there).
template <int K, int TB, int NB>
struct gmm_smem {
  static constexpr int NC = K * POCS_NMOM;
  static constexpr int RB = TB / 16;                                 // rows of sums per task: one per 16-lane DPP row
  alignas(16) pocs_tables tab;                                       // 12 KB log / sector tables, staged once per block
  alignas(16) double obs[POCS_MAX_OBSTACLES * POCS_OBS_STRIDE];      // obstacle table, staged once per block
  alignas(16) double keep[NB][POCS_MAX_OBSTACLES * POCS_OBS_STRIDE]; // ... culled for a task
  alignas(16) double par[NB][K * POCS_PARAM_STRIDE];                 // sampler parameters of the task's (run, waypoint)
  double red[NB][RB][NC];                                            // the task's row sums; staging area of its closer
  double adv[POCS_ADV_SCRATCH(K)];                                   // mixture advance: inputs and outputs
  double spec[POCS_SPEC_SCRATCH(K)];
  double tot[NC];
  int nkeep[NB];
  int last;                 // k_gmm_step: this block drew the last ticket
  // k_gmm_run, the block's task pipeline: buffer b = n % NB holds the block's n-th task
  unsigned long long seed[NB];      // what every wave needs of it, worked out once by the loader: the run's seed,
  long long c_begin[NB], c_end[NB]; // its chunk range,
  int tw[NB], tr[NB], tslot[NB];    // waypoint, run, slice
  unsigned task_id[NB];     // its number in the launch's queue (>= total: no more tasks, leave)
  unsigned seq[NB];         // n + 1 once wave 0 has staged it (parameters, culled table)
  unsigned done[NB];        // waves that have sampled their share of it and drained their stores
  unsigned freed[NB];       // n + 1 once its closing wave is through with the buffer
  unsigned lock;            // the advance scratch above is one wave's at a time
  unsigned quit;            // a bounded wait expired somewhere: everybody leaves
};

// Mixture bookkeeping of waypoint `w` (pocs_gmm_advance_component / pocs_gmm_normalise): every input
// (state[w-1], the reduced moments of w-1, the chain record of step w-1, the sensor) is first brought
// to LDS in ONE round trip, lanes < K of one wave then take one component each (truncated mean /
// covariance, EKF predict + update, Cholesky); lane 0 normalises, draws the component counts, and the
// wave writes state[w] / param[w] back write-through.  Run by a whole block (k_gmm_advance,
// k_gmm_step: a lane of a second wave draws the counts meanwhile, on the premise -- checked -- that
// no factorisation fails) or by a single wave (k_gmm_run); same functions, same results.
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

// n values src(0) .. src(n-1) -> stage[0 .. n-1] by `nthreads` threads, U loads in flight per thread:
// the loads of a batch are all issued before the first of them is waited for (a plain copy loop waits
// for every load before it issues the next one: one memory round trip per element and thread).
template <int U, typename Src>
__device__ __forceinline__ void stage_batched(double* stage, const int n, const int tid, const int nthreads, Src src) {
  for (int i0 = tid; i0 < n; i0 += nthreads * U) {
    double v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = i0 + u * nthreads; v[u] = (i < n) ? src(i) : 0.0; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const int i = i0 + u * nthreads; if (i < n) stage[i] = v[u]; }
  }
}

// `nthreads` threads (tid < nthreads).  mom_in_lds: l_mom already holds the moments of w-1 (the caller
// has just reduced them); otherwise they are read from a.moments (own launch: after the caller's
// all-reduce).  state[w-1] may have been written by another block of THIS launch: L1-bypassing loads.
// One batch of loads for everything (the scratch is laid out l_prev | l_mom | l_ch | l_sen).
__device__ __forceinline__ void advance_stage(const pocs_gmm_launch& a, int K, int w, int r, double* scratch,
                                              bool mom_in_lds, int tid, int nthreads) {
  const adv_ptrs p = advance_ptrs(a, K, w, r, scratch);
  constexpr int SEN = (int)(sizeof(pocs_sensor) / sizeof(double));
  const int ss = p.ss, NC = p.NC;
  const bool load_mom = w > 0 && !mom_in_lds;
  // index space: [0, ss) state | [ss, ss + CH + SEN) chain record, sensor | then (only if wanted) the moments;
  // l_mom is NOT touched when the caller has put the moments there
  const int n = ss + POCS_CHAIN_STRIDE + SEN + (load_mom ? NC : 0);
  for (int i0 = tid; i0 < n; i0 += nthreads * 4) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * nthreads;
      v[u] = 0.0;
      if (i < ss) v[u] = load_wt(&p.g_prev[i]);
      else if (i < ss + POCS_CHAIN_STRIDE) v[u] = p.g_ch[i - ss];
      else if (i < ss + POCS_CHAIN_STRIDE + SEN) v[u] = p.g_sen[i - ss - POCS_CHAIN_STRIDE];
      else if (i < n) v[u] = p.g_mom[i - ss - POCS_CHAIN_STRIDE - SEN];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * nthreads;
      if (i < ss) p.l_prev[i] = v[u];
      else if (i < ss + POCS_CHAIN_STRIDE + SEN) p.l_ch[i - ss] = v[u];              // l_ch | l_sen are contiguous
      else if (i < n) p.l_mom[i - ss - POCS_CHAIN_STRIDE - SEN] = v[u];
    }
  }
}

// one wave, after advance_stage (+ barrier): one component per lane
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

// the wave of advance_components, after it (+ barrier): weights, component counts (the speculated ones
// if there are any and their premise held), write-through stores of state[w] / param[w], drained.
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
#if defined(POCS_STEP_STAMPS)      // k_gmm_step's closer only (w >= 1 there): where the advance spends its time
  unsigned long long* const dbg_ = (mom_in_lds && gridDim.y > 1 + 0 * K) || mom_in_lds
      ? a.dbg + (((size_t)(w - 1) * a.nruns + r) * a.slices + blockIdx.x) * 32 : nullptr;
#define POCS_ADV_STAMP(i) do { if (dbg_ && tid == 0) dbg_[i] = wall_clock64(); } while (0)
#else
#define POCS_ADV_STAMP(i) do { } while (0)
#endif
  advance_stage(a, K, w, r, adv, mom_in_lds, tid, nthreads);
  __syncthreads();
  POCS_ADV_STAMP(26);
  if (tid < 64) advance_components(a, K, w, r, tid, adv);
  else if (tid == 64 && w > 0) speculate_counts(a, K, w, r, adv, spec);
  POCS_ADV_STAMP(27);
  __syncthreads();
  POCS_ADV_STAMP(28);
  if (tid < 64) advance_finish(a, K, w, r, tid, adv, w > 0 ? spec : nullptr);
}
// ... and by ONE wave (all 64 lanes call it), the moments of w-1 already in l_mom.
__device__ __forceinline__ void advance_wave(const pocs_gmm_launch& a, int K, int w, int r, double* adv, int lane) {
  advance_stage(a, K, w, r, adv, true, lane, 64);
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
  advance_components(a, K, w, r, lane, adv);
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
  advance_finish(a, K, w, r, lane, adv, nullptr);
}

__global__ __launch_bounds__(128) void k_gmm_advance(pocs_gmm_launch a, int K) {
  __shared__ double s_adv[POCS_ADV_SCRATCH(POCS_MAX_GAUSSIANS)];
  __shared__ double s_spec[POCS_SPEC_SCRATCH(POCS_MAX_GAUSSIANS)];
  advance_block(a, K, a.waypoint, blockIdx.x, s_adv, s_spec, false, threadIdx.x, 128);      // one block per run
}

// ---------------------------------------------------------------------------------------------
// Sharded over the GPUs of a node: the moments of waypoint w of this rank's samples (moments[w][r],
// left by k_gmm_step) -> the moments of the whole mixture, in ONE hop over xGMI instead of a ring
// (11 K doubles per run are latency, not bandwidth), and straight on to the mixture
// of waypoint w+1 -- exchange + advance in one small launch (one block per run) between two sampling
// launches.  Every rank owns a buffer that all ranks have mapped; rank q writes its row into slot q
// of EVERY buffer (system-scope stores: peers sit across xGMI), drains, meets, writes the slot's flag
// = this waypoint's epoch; then waits for the world's flags in its OWN buffer and adds the slots in
// rank order -- every rank the same sum, bit for bit, whatever the arrival order.  Slots alternate
// with the waypoint's parity: a rank can only be one exchange ahead of another.  The wait is bounded
// (30 s, once per call: later exchanges of a call that has given up return at once): on expiry the
// call's give-up word is set and the host reports POCS_E_DEVICE.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double* xchg_row(double* buf, int parity, int src, int r) {
  return buf + (((size_t)parity * POCS_XCHG_MAX_WORLD + src) * POCS_XCHG_MAX_RUNS + r) * POCS_XCHG_MAX_NC;
}
__device__ __forceinline__ unsigned long long* xchg_flag(double* buf, int parity, int src, int r) {
  return reinterpret_cast<unsigned long long*>(buf + POCS_XCHG_DATA_DOUBLES) +
         ((size_t)parity * POCS_XCHG_MAX_WORLD + src) * POCS_XCHG_MAX_RUNS + r;
}
// The exchange itself, by the `nthreads` threads of one block for run r at waypoint w: `mine` (NC doubles,
// global or LDS) -> slot `rank` of every rank's buffer; wait for the world's rows; the sum in rank order
// -> a.moments[w][r] and l_mom (LDS; may be `mine`).  s_ok: one int of LDS.  false = gave up.
__device__ __forceinline__ bool gmm_exchange_rows(const pocs_gmm_launch& a, const pocs_xchg_dev& x, const int K, const int w, const int r,
                                                  const double* mine, double* l_mom, const int tid, const int nthreads, int* s_ok) {
  const int NC = K * POCS_NMOM, parity = w & 1;
  for (int i = tid; i < NC * x.world; i += nthreads) {
    const int q = i / NC, c = i - q * NC;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(xchg_row(x.buf[q], parity, x.rank, r) + c),
                       (unsigned long long)__double_as_longlong(mine[c]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // the rows are out before the flags: they were stored write-through at system scope, and every storing
  // wave waits for its stores here.  (NOT a system-scope release fence: that writes back the whole L2, and
  // in the tail of a sampling launch the L2 is full of samples on their way out -- measured +45 us per
  // 64-run launch.)
  drain_stores();
  __syncthreads();
  if (tid < x.world)
    __hip_atomic_store(xchg_flag(x.buf[tid], parity, x.rank, r), x.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // every rank's row of this waypoint has landed in MY buffer?
  if (tid == 0) *s_ok = 1;
  __syncthreads();
  if (tid < x.world) {
    const unsigned long long* f = xchg_flag(x.buf[x.rank], parity, tid, r);
    const unsigned long long t0 = wall_clock64();
    unsigned polls = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != x.epoch) {
      __builtin_amdgcn_s_sleep(8);
      if ((++polls & 255u) == 0u && wall_clock64() - t0 > 3000000000ull) {      // 30 s: ranks of a cold node start seconds apart
        __hip_atomic_store(&a.sync[POCS_SYNC_ABORT], 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_ok = 0;
        break;
      }
    }
  }
  if (tid < 64) {                                          // ONE wave, the one that polled: drop this XCD's stale lines
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // system scope
    drain_stores();                                        // the invalidate completes before the barrier releases the readers
  }
  __syncthreads();
  if (!*s_ok) return false;
  // the slots of my buffer, added in rank order
  for (int c = tid; c < NC; c += nthreads) {
    double tot = 0.0;
    for (int q = 0; q < x.world; ++q)
      tot += __longlong_as_double((long long)__hip_atomic_load(
          reinterpret_cast<const unsigned long long*>(xchg_row(x.buf[x.rank], parity, q, r) + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    a.moments[((size_t)w * a.nruns + r) * NC + c] = tot;    // the mixture's moments replace this shard's
    l_mom[c] = tot;
  }
  return true;
}
__global__ __launch_bounds__(128) void k_gmm_exchange(pocs_gmm_launch a, pocs_xchg_dev x, int K) {
  __shared__ double s_adv[POCS_ADV_SCRATCH(POCS_MAX_GAUSSIANS)];
  __shared__ double s_spec[POCS_SPEC_SCRATCH(POCS_MAX_GAUSSIANS)];
  __shared__ int s_ok;
  const int tid = threadIdx.x, r = blockIdx.x, w = a.waypoint, NC = K * POCS_NMOM;
  // an earlier exchange of this call gave up: do not wait another 30 s per waypoint, the call is lost
  if (__hip_atomic_load(&a.sync[POCS_SYNC_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
  double* const l_mom = advance_ptrs(a, K, w + 1, r, s_adv).l_mom;
  if (!gmm_exchange_rows(a, x, K, w, r, a.moments + ((size_t)w * a.nruns + r) * NC, l_mom, tid, 128, &s_ok)) return;
  if (w + 1 < a.W) advance_block(a, K, w + 1, r, s_adv, s_spec, true, tid, 128);     // starts with a barrier after staging
}

// A wave leaves component `k`: its 16-lane row sums of (nFree, 9 sums) are ADDED to the task's LDS
// rows of that component -- every 16-lane row of threads owns one row of sums, a wave meets a component
// once per task, the add only matters for a component it never touched (+ 0) -- and the thread-private
// sums restart.  Column 1 of a row (nColl) stays 0: the collisions of a component are what is left of
// its block of samples (gmm_close_sums).
template <int NC>
__device__ __forceinline__ void flush_component(double (*s_red)[NC], int k, double (&acc)[10], int tid) {
  const int row = tid >> 4;
  const bool writer = (tid & 15) == 0;
  double* dst = &s_red[row][k * POCS_NMOM];
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    const double v = row_sum(acc[j]);
    if (writer) dst[j == 0 ? 0 : 1 + j] += v;
    acc[j] = 0.0;
  }
}

// Once per block: the log / sector tables and the obstacle table -> LDS.
template <int K, int TB, int NB>
__device__ __forceinline__ void gmm_stage_static(const pocs_gmm_launch& a, gmm_smem<K, TB, NB>& sm) {
  stage_tables(a.tables, &sm.tab);
  for (int j = threadIdx.x; j < a.M * POCS_OBS_STRIDE; j += TB) sm.obs[j] = a.env->obs[j];
}

// Cull the obstacle table against the bounding box of the mixture staged in par[buf] (ONE wave, all 64
// lanes).  A Box-Muller normal is bounded: u >= 2^-32 gives |z| <= sqrt(64 ln 2) < 6.661
// (pocs_normal_pair_w2; 6.67 leaves 0.1 % for the rounding of radius * cos), so every pose the task can
// draw lies within mean_k +- 6.67 (|L00|, |L10|+|L11|) of some component; an obstacle whose inflated
// box (the broad phase of pocs_box_hit) misses that region is rejected by the broad phase for every
// sample, so dropping it here changes no flag.
//
// The same bound on the heading makes the broad phase of the kept records tighter than the table's: the
// table inflates an obstacle's box by the footprint's bounding RADIUS (any heading); a task whose
// headings all lie in [t_lo, t_hi] needs only the footprint's largest half-extent along world x and
// along world y over that range (two convex sets that touch overlap in every projection).  Where the
// robot's heading is known to a fraction of a radian -- most of a plan -- far fewer poses reach the
// narrow phase, and none that could touch is lost: the flags do not change.
//   (pocs_footprint_extent, pocs_collide.h: host + device, checked on the CPU against a dense scan)
template <int K, int TB, int NB>
__device__ __forceinline__ void gmm_cull(const pocs_gmm_launch& a, gmm_smem<K, TB, NB>& sm, const int buf, const int lane) {
  const pocs_footprint fp = a.fp;
  const int M = a.M;
  double xlo = 1e300, xhi = -1e300, ylo = 1e300, yhi = -1e300, tlo = 1e300, thi = -1e300;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double* p = &sm.par[buf][k * POCS_PARAM_STRIDE];
    const double ex = 6.67 * fabs(p[3]), ey = 6.67 * (fabs(p[4]) + fabs(p[5])), et = 6.67 * (fabs(p[6]) + fabs(p[7]) + fabs(p[8]));
    xlo = fmin(xlo, p[0] - ex); xhi = fmax(xhi, p[0] + ex);
    ylo = fmin(ylo, p[1] - ey); yhi = fmax(yhi, p[1] + ey);
    tlo = fmin(tlo, p[2] - et); thi = fmax(thi, p[2] + et);
  }
  const double pad = sqrt(fp.dx * fp.dx + fp.dy * fp.dy) + 1e-6;   // footprint centre vs base
  xlo -= pad; xhi += pad; ylo -= pad; yhi += pad;
  tlo -= 1e-9 * (1.0 + fabs(tlo)); thi += 1e-9 * (1.0 + fabs(thi));
  const double HALF_PI = 1.57079632679489661923;
  const double ext_x = pocs_footprint_extent(fp.hx, fp.hy, tlo, thi);
  const double ext_y = pocs_footprint_extent(fp.hx, fp.hy, tlo - HALF_PI, thi - HALF_PI);
  bool keep = false;
  double bx = 0.0, by = 0.0;
  if (lane < M) {
    const double* o = &sm.obs[lane * POCS_OBS_STRIDE];
    // the obstacle's own world box (as pocs_prepare_obstacle) + the footprint's extents for this task
    bx = fmin(o[6], fma(o[4], fabs(o[2]), o[5] * fabs(o[3])) * (1.0 + 1e-12) + ext_x);
    by = fmin(o[7], fma(o[4], fabs(o[3]), o[5] * fabs(o[2])) * (1.0 + 1e-12) + ext_y);
    keep = !(o[0] - bx > xhi || o[0] + bx < xlo || o[1] - by > yhi || o[1] + by < ylo);
  }
  const unsigned long long mask = __ballot(keep);
  if (keep) {
    const int pos = __popcll(mask & ((1ull << lane) - 1ull));
#pragma unroll
    for (int j = 0; j < 6; ++j) sm.keep[buf][pos * POCS_OBS_STRIDE + j] = sm.obs[lane * POCS_OBS_STRIDE + j];
    sm.keep[buf][pos * POCS_OBS_STRIDE + 6] = bx;
    sm.keep[buf][pos * POCS_OBS_STRIDE + 7] = by;
  }
  if (lane == 0) sm.nkeep[buf] = __popcll(mask);
}

// ---------------------------------------------------------------------------------------------
// THE BODY of a task = slice `slot` (of a.slices) of run r at waypoint w, as every thread of the block
// runs it: GM_Model::sampleNPoints (GM_Model.h:83-116) + checkMatrixCollisions (MCSimulator.h:241-253)
// + the moment sums (:592-611), fused, one PAIR of samples per thread and iteration; ends with the
// thread's sums folded into the task's LDS rows (each 16-lane row of threads owns one row: no
// barrier in here).  s_par / s_keep / nkeep: the task's staged parameters and culled obstacle table.
// WT: write-through sample stores (k_gmm_run).  What a thread computes depends on (r, w, slot,
// a.slices, a.chunks, its tid) only -- not on the kernel, the block, or when: k_gmm_step and
// k_gmm_run produce bitwise the same rows.  seed = the run's; [c_begin, c_end) = the slice's chunks
// (gmm_slice_chunks), both wave-uniform.
// ---------------------------------------------------------------------------------------------
template <int K, bool STORE, bool WT, int TB>
__device__ __forceinline__ void gmm_body(const pocs_gmm_launch& a, const pocs_tables* s_tab, const double* s_par,
                                         const double* s_keep, const int nkeep, double (*s_red)[K * POCS_NMOM],
                                         const int w, const int r, uint64_t seed, const long long c_begin, const long long c_end
#if defined(POCS_TASK_STAMPS)
                                         , unsigned long long (&st_)[12], unsigned long long& last_
#endif
                                         ) {
  constexpr int NC = K * POCS_NMOM;
  const int tid = threadIdx.x;
  const pocs_footprint fp = a.fp;
  seed = (uint64_t)uniform64((long long)seed);       // scalar registers, provably
  // Samples come in component blocks and a thread's sample indices only grow, so a wave works
  // through the components in order: ONE set of sums per thread (the wave's current component),
  // folded into the task's LDS rows when the wave moves on to the next component.
  //   acc[0] = survivors, acc[1..9] = sums of x, y, t, xx, xy, xt, yy, yt, tt over them
  double acc[10];
  int kcur = 0;                                      // wave-uniform
#pragma unroll
  for (int j = 0; j < 10; ++j) acc[j] = 0.0;
  for (int c = tid & 15; c < NC; c += 16) s_red[tid >> 4][c] = 0.0;      // this row of threads' own row of sums

  // a.first is even (checked by the host), so local sample 2*lp is global sample first + 2*lp.
  const long long npairs = (a.count + 1) >> 1;
  const uint64_t pair0 = (uint64_t)(a.first >> 1);
  const double first_d = (double)a.first;
  double cumn[K > 1 ? K - 1 : 1];                   // cumulative component counts (wave-uniform)
#pragma unroll
  for (int j = 0; j < K - 1; ++j) cumn[j] = s_par[j * POCS_PARAM_STRIDE + 9];
  // (Keeping the wave's component parameters in scalar registers instead of reading them at one LDS
  // address spills scalar registers: measured 5 % slower at K = 3.)
  const long long wave_first = 2 * (long long)(__builtin_amdgcn_readfirstlane(tid >> 6) * 64);
  const long long g_end = a.first + a.count;
  long long seg_end = 0;
  int kw = 0;
  // The loop counter is wave-uniform (SGPRs) and the lane adds its tid: the store addresses are a
  // scalar base per iteration plus a constant 16*tid, no per-lane 64-bit address arithmetic.
  double* const xr = a.x + (size_t)r * a.sample_stride;          // this run's slice (sample_stride is even)
  double* const yr = a.y + (size_t)r * a.sample_stride;
  double* const tr = a.th + (size_t)r * a.sample_stride;
  int16_t* const fr = a.flags + (size_t)r * a.sample_stride;
#if !defined(POCS_NO_PRIO_ROTATION)
  // The (up to) four waves of a SIMD -- two of this block, two of the co-resident one -- are arbitrated
  // by priority, then AGE: left alone, the oldest wave of a SIMD runs ~1.7 x faster than the youngest for
  // the whole launch.  Rotating the priority with the iteration gives every wave the same share.
  // slot = which of the block's waves on this SIMD: wave v runs on SIMD v mod 4, so waves v and v + 4 share
  // one (HW_ID stamps of a diagnostic build, tools/step_stamps.sh).  The second block of a CU is (observed,
  // speed only) the one dispatched 256 blocks later.  What the rotation does NOT achieve (same stamps): the
  // older of a SIMD's waves still win -- a block's waves 4-7 end 3 % after its waves 0-3, the blocks
  // dispatched second 19 % after the first.  Rotating by the wall clock instead of the iteration count (equal
  // TIME at each level, no two waves of a SIMD ever tied) measured worse at 2.5 and 5 us per level, +0.8 % at 10.
  const int prio_slot = (TB >= 512 ? __builtin_amdgcn_readfirstlane((tid >> 6) >> 2) : 0) +
                        (TB >= 512 ? 2 : 1) * (int)(((blockIdx.x + gridDim.x * blockIdx.y) >> 8) & 3u);
  int prio_it = prio_slot;
#endif
  POCS_STAMP(2);
  // ONE iteration = 2 * TB samples, one pair per thread.  WHOLE (compile time): the wave's 128 samples lie
  // inside component block kw and inside the shard -- every lane live, both samples of its pair exist,
  // the component is the scalar kw == kcur.  Otherwise: the general case (a block boundary inside the
  // wave, the shard's last chunk), every decision per lane.  Same arithmetic per sample either way.
  auto iteration = [&](auto whole_tag, const long long base) __attribute__((always_inline)) {
    constexpr bool WHOLE = decltype(whole_tag)::value;
#if !defined(POCS_NO_PRIO_ROTATION)
    switch (prio_it++ & 3) {                       // s_setprio takes an immediate
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
#endif
    const long long lp = base + tid;
    const bool live = WHOLE || lp < npairs;        // a lane past the end computes, masked: the row sums below need every lane
    double zz[2][3];
    uint32_t spare[2];
#if defined(POCS_ABLATE_RNG)          // timing-only builds (tools/ablate.sh): outputs are wrong
    for (int h = 0; h < 2; ++h) { zz[h][0] = (double)(lp & 7) * 0.1; zz[h][1] = (double)(lp & 3) * 0.1; zz[h][2] = 0.05; spare[h] = (uint32_t)lp * 2654435761u; }
#elif defined(POCS_ABLATE_PHILOX)      // Box-Muller kept, its words from a three-instruction hash: what Philox costs in situ
    { const uint32_t q = (uint32_t)(pair0 + lp) * 2654435761u ^ (uint32_t)seed ^ ((uint32_t)w << 20);
      const uint32_t a0 = q * 0x9E3779B1u, a1 = (q ^ 0x85EBCA6Bu) * 0xC2B2AE35u, a2 = (q + 0x27D4EB2Fu) * 0x165667B1u;
      pocs_normal_pair_w2(a0, a1, s_tab, &zz[0][0], &zz[0][1]);
      pocs_normal_pair_w2(a2, a0 ^ a1, s_tab, &zz[0][2], &zz[1][0]);
      pocs_normal_pair_w2(a1 ^ a2, a0 + a2, s_tab, &zz[1][1], &zz[1][2]);
      spare[0] = a0; spare[1] = a1; }
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
    pocs_normal3_pair(seed_it, pair0 + (uint64_t)lp, (uint32_t)w, POCS_STREAM_GMM, s_tab, zz[0], zz[1], &spare[0], &spare[1]);
#endif
    const long long i0 = 2 * lp;
    const bool two = WHOLE || (live && (i0 + 1) < a.count);  // false only for the last sample of an odd shard
    double xs[2], ys[2], ts[2];
    bool hits[2];
    int ks[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int k = kw;                                   // WHOLE: one LDS address for the wave, broadcast reads
      if (!WHOLE) {
        // component of the sample (GM_Model.h:87-107: counts[k] samples per component, one block
        // after the other): the first component whose cumulative count exceeds the global index
        const double gidx = (first_d + (double)i0) + (double)h;   // exact: < 2^53
        k = 0;
#pragma unroll
        for (int j = 0; j < K - 1; ++j) k += (cumn[j] <= gidx) ? 1 : 0;
      }
      const double* p = &s_par[k * POCS_PARAM_STRIDE];
      // mvnrnd (glue_mvnrnd_meat.hpp:134-145): chol_lower * z + mean
      xs[h] = fma(p[3], zz[h][0], p[0]);
      ys[h] = fma(p[5], zz[h][1], fma(p[4], zz[h][0], p[1]));
      ts[h] = fma(p[8], zz[h][2], fma(p[7], zz[h][1], fma(p[6], zz[h][0], p[2])));
      ks[h] = k;
#if defined(POCS_ABLATE_COLLIDE)
      hits[h] = xs[h] > ts[h];
#else
      hits[h] = pocs_pose_collides(xs[h], ys[h], ts[h], &fp, s_keep, nkeep, s_tab);
#endif
    }
#if defined(POCS_ABLATE_MOMENTS)
    acc[1] += xs[0] + ys[0] + ts[0] + xs[1]; acc[0] += (hits[0] || (two && ks[1] == 0)) ? 0.0 : 1.0;
#else
    // T1 sums over the collision-free samples of the component being accumulated:
    //   acc += (1, x, y, t, x x, x y, x t, y y, y t, t t)      (the products inside the fma).
    if (WHOLE) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (!hits[h]) {                             // the few lanes that collided sit this out
          const double x = xs[h], y = ys[h], t = ts[h];
          acc[0] += 1.0;
          acc[1] += x; acc[2] += y; acc[3] += t;
          acc[4] = fma(x, x, acc[4]); acc[5] = fma(x, y, acc[5]); acc[6] = fma(x, t, acc[6]);
          acc[7] = fma(y, y, acc[7]); acc[8] = fma(y, t, acc[8]); acc[9] = fma(t, t, acc[9]);
        }
      }
    } else {
      // The components present in the wave are visited in increasing order (scalar loop), the previous
      // component's sums being flushed to the LDS rows first; with ind = 1.0 for a surviving sample of the
      // component and 0.0 otherwise, (xm, ym, tm) = ind * (x, y, t) enter the sums -- a sample that does not
      // count adds +-0 to every one of them, which is why the WHOLE form above gives the same bits.
      // Sample indices grow with the lane: the wave's first LIVE lane holds its first component, lane 63 its last.
      const unsigned long long live_mask = __ballot(live);
      const int klo = live_mask ? __builtin_amdgcn_readlane(ks[0], (int)__builtin_ctzll(live_mask)) : K;
      const int khi = __ballot(two) == ~0ull ? __builtin_amdgcn_readlane(ks[1], 63) : K - 1;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        if (kk < klo || kk > khi) continue;                                 // scalar compares
        if (kk != kcur) { flush_component<NC>(s_red, kcur, acc, tid); kcur = kk; }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool sel = (h == 0 ? live : two) && ks[h] == kk;
          const double ind = (sel && !hits[h]) ? 1.0 : 0.0;
          const double xm = ind * xs[h], ym = ind * ys[h], tm = ind * ts[h];
          acc[0] += ind;
          acc[1] += xm; acc[2] += ym; acc[3] += tm;
          acc[4] = fma(xm, xs[h], acc[4]); acc[5] = fma(xm, ys[h], acc[5]); acc[6] = fma(xm, ts[h], acc[6]);
          acc[7] = fma(ym, ys[h], acc[7]); acc[8] = fma(ym, ts[h], acc[8]); acc[9] = fma(tm, ts[h], acc[9]);
        }
      }
    }
#endif
    if (STORE && live) {
      // Both poses of the pair leave together.  For the last sample of an odd shard the second slot is
      // the pair's unused twin: it lands in the padding element of the run's slice (sample_stride >=
      // count + 1 then) and is never read back.  Written once, never re-read by the kernels.
      // (Issuing the three pose stores BEFORE the footprint test, so that they drain under it, was
      // measured equal: r02, 507 vs 496 us per 64-run launch.)
      const size_t ub = 2 * (size_t)base;
      const int fl = (hits[0] ? 1 : 0) | ((two && hits[1]) ? 0x10000 : 0);
      if (WT) {                                    // k_gmm_run: write-through, see store16_wt
        store16_wt(xr + ub, 16u * (unsigned)tid, (v2d){xs[0], xs[1]});
        store16_wt(yr + ub, 16u * (unsigned)tid, (v2d){ys[0], ys[1]});
        store16_wt(tr + ub, 16u * (unsigned)tid, (v2d){ts[0], ts[1]});
        store4_wt(fr + ub, 4u * (unsigned)tid, fl);
      } else {                                     // a launch per waypoint: non-temporal, past the caches
        store16_nt(xr + ub, 16u * (unsigned)tid, (v2d){xs[0], xs[1]});
        store16_nt(yr + ub, 16u * (unsigned)tid, (v2d){ys[0], ys[1]});
        store16_nt(tr + ub, 16u * (unsigned)tid, (v2d){ts[0], ts[1]});
        store4_nt(fr + ub, 4u * (unsigned)tid, fl);
      }
    }
  };
  // A wave's 128 samples of an iteration nearly always lie inside ONE component block and inside the shard:
  // those iterations run in the inner loop below, which knows nothing of the general case (no per-lane
  // index compares, no masks, no flush; the accumulators stay where they are).  The wave's position is
  // looked up again (scalar) whenever the next iteration is not of that kind.
  //   seg_end = global index up to which (exclusive) whole waves belong to component kw and exist
  const long long end = c_end * TB;
  for (long long base = c_begin * TB; base < end;) {
    long long g0 = a.first + 2 * base + wave_first;                         // global index of the wave's first sample
    {
      int kk = 0;
#pragma unroll
      for (int j = 0; j < K - 1; ++j) kk += (cumn[j] <= (double)g0) ? 1 : 0;
      kw = __builtin_amdgcn_readfirstlane(kk);
      const long long blk_end = kw < K - 1 ? (long long)s_par[kw * POCS_PARAM_STRIDE + 9] : g_end;
      seg_end = uniform64(blk_end < g_end ? blk_end : g_end);
    }
    if (g0 + 128 <= seg_end) {
      if (kw != kcur) { flush_component<NC>(s_red, kcur, acc, tid); kcur = kw; }
      do {
        iteration(std::true_type{}, base);
        base += TB;
        g0 += 2 * TB;
      } while (base < end && g0 + 128 <= seg_end);
    } else {
      iteration(std::false_type{}, base);
      base += TB;
    }
  }
#if !defined(POCS_NO_PRIO_ROTATION)
  // Everything between two bodies -- flush, drain, closing a task, adding a run's rows, advancing its
  // mixture, staging the next task -- is short and other waves (or blocks) wait for it: it runs at the
  // TOP priority, or the sampling waves of the same SIMD (priority 0..3 in turn) leave it the crumbs.
  __builtin_amdgcn_s_setprio(3);
#endif
  POCS_STAMP(3);
  flush_component<NC>(s_red, kcur, acc, tid);       // the last component's sums -> the LDS rows
  POCS_STAMP(4);
}

// chunks [begin, end) of slice `slot`
__device__ __forceinline__ void gmm_slice_chunks(const pocs_gmm_launch& a, const int slot, long long* begin, long long* end) {
  *begin = ((long long)slot * a.chunks) / a.slices;
  *end = ((long long)(slot + 1) * a.chunks) / a.slices;
}

// Column c of the task's partial row: the RB rows of sums added in row order.
template <int NC, int RB>
__device__ __forceinline__ double gmm_row_total(const double (*s_red)[NC], const int c) {
  double v = s_red[0][c];
#pragma unroll 8
  for (int q = 1; q < RB; ++q) v += s_red[q][c];
  return v;
}

// The closer of (r, w) -- the block (k_gmm_step) or wave (k_gmm_run) that drew the run's last ticket,
// behind its acquire -- adds the a.slices partial rows of the run in a FIXED order that does not depend
// on who adds them: sixteen interleaved partial sums per column, P_g = row g + row (g + 16) + row (g + 32)
// + ... in that order, then P_0 + P_1 + ... + P_15 in that order.  (Up to 16 slices -- 32 runs per
// launch and more -- that is plain slice order.)  Every (g, column) is one work item: its loads are
// L1-bypassing and independent, so the `nthreads` threads have the whole table in flight at once --
// ONE memory round trip instead of one per 32 rows, which is what a lone run's 256 slices used to
// cost (7 us of a 31 us launch).  `stage`: >= 16 * NC doubles of LDS.  `sync` = the barrier of those
// threads.  The collisions of a component are what is left of its block of this shard's samples:
// nColl_k = n_k - nFree_k, with [cum_{k-1}, cum_k) the component's global sample range (par[k][9],
// cum_{K-1} = n_total).  Result: tot[c], and moments[w][r][c] in global memory (it leaves the launch at
// the kernel boundary).
template <int K, int RB, typename Sync>
__device__ __forceinline__ void gmm_close_sums(const pocs_gmm_launch& a, const int w, const int r, const double* s_par,
                                               double* stage, double* tot, const int tid, const int nthreads, Sync sync) {
  constexpr int NC = K * POCS_NMOM, G = 16;
  static_assert(RB >= G, "the staging rows hold the sixteen partial sums");
  const int S = a.slices;
  const double* src = a.partial + (size_t)r * S * NC;
  for (int i = tid; i < G * NC; i += nthreads) {
    const int g = i / NC, c = i - g * NC;
    double v = 0.0;
    int q = g;
    for (; q + 3 * G < S; q += 4 * G) {              // four rows of the item in flight at a time, added in row order
      const double v0 = load_wt(&src[(size_t)q * NC + c]), v1 = load_wt(&src[(size_t)(q + G) * NC + c]);
      const double v2 = load_wt(&src[(size_t)(q + 2 * G) * NC + c]), v3 = load_wt(&src[(size_t)(q + 3 * G) * NC + c]);
      v += v0; v += v1; v += v2; v += v3;
    }
    for (; q < S; q += G) v += load_wt(&src[(size_t)q * NC + c]);
    stage[i] = v;
  }
  sync();
  for (int c = tid; c < NC; c += nthreads) {
    double v = stage[c];
    const int ng = S < G ? S : G;
    for (int g = 1; g < ng; ++g) v += stage[g * NC + c];
    tot[c] = v;
  }
  sync();
  const double lo = (double)a.first, hi = (double)(a.first + a.count);
  for (int k = tid; k < K; k += nthreads) {
    const double c0 = (k == 0) ? 0.0 : s_par[(k - 1) * POCS_PARAM_STRIDE + 9];
    const double c1 = (k == K - 1) ? (double)a.n_total : s_par[k * POCS_PARAM_STRIDE + 9];
    const double n_k = fmax(0.0, fmin(c1, hi) - fmax(c0, lo));
    tot[k * POCS_NMOM + 1] = n_k - tot[k * POCS_NMOM];
  }
  sync();
  for (int c = tid; c < NC; c += nthreads) a.moments[((size_t)w * a.nruns + r) * NC + c] = tot[c];
}

// One waypoint as its own launch: grid = (slices, runs), block (j, r) = task (a.waypoint, r, j).  The
// per-waypoint form, for a caller that exchanges the moments between waypoints (sharded over GPUs) and
// for calls with too few runs to keep k_gmm_run's pipeline full.
//   head  sampler parameters of (r, w) -> LDS, exact culling of the obstacle table;
//   body  gmm_body;
//   tail  the task's rows -> ONE write-through partial row (r, slot) -> every storing wave drains ->
//         the block meets -> ticket (r, w); the block that draws the last one acquires, adds the
//         run's partial rows and (one GPU) advances the mixture to the next waypoint.
template <int K, bool STORE, int TB>
__global__ __launch_bounds__(TB, POCS_GMM_BLOCKS_PER_CU * TB / 256) void k_gmm_step(pocs_gmm_launch a) {
  typedef gmm_smem<K, TB, 1> smem_t;
  constexpr int NC = smem_t::NC, RB = smem_t::RB;
  __shared__ smem_t sm;
  const int tid = threadIdx.x;
  const int w = a.waypoint, r = blockIdx.y, slot = blockIdx.x;
#if defined(POCS_STEP_STAMPS)
  unsigned long long* const dbg_ = a.dbg + (((size_t)w * a.nruns + r) * a.slices + slot) * 32;
#define POCS_STEP_STAMP(i) do { if (tid == 0) dbg_[i] = wall_clock64(); } while (0)
#else
#define POCS_STEP_STAMP(i) do { } while (0)
#endif
  POCS_STEP_STAMP(0);
  gmm_stage_static(a, sm);
  for (int j = tid; j < K * POCS_PARAM_STRIDE; j += TB)
    sm.par[0][j] = load_wt(&a.param[((size_t)r * a.W + w) * (K * POCS_PARAM_STRIDE) + j]);
  __syncthreads();
  if (tid < 64) gmm_cull(a, sm, 0, tid);
  __syncthreads();
  long long c_begin, c_end;
  gmm_slice_chunks(a, slot, &c_begin, &c_end);
  c_begin = uniform64(c_begin); c_end = uniform64(c_end);      // (64-bit division runs on the vector unit)
  POCS_STEP_STAMP(1);
#if defined(POCS_TASK_STAMPS)
  unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = 0;
  gmm_body<K, STORE, false, TB>(a, &sm.tab, sm.par[0], sm.keep[0], __builtin_amdgcn_readfirstlane(sm.nkeep[0]), sm.red[0], w, r,
                                a.hdr[r].seed, c_begin, c_end, st_, last_);
#else
  gmm_body<K, STORE, false, TB>(a, &sm.tab, sm.par[0], sm.keep[0], __builtin_amdgcn_readfirstlane(sm.nkeep[0]), sm.red[0], w, r,
                                a.hdr[r].seed, c_begin, c_end);
#endif
  POCS_STEP_STAMP(2);
#if defined(POCS_STEP_STAMPS)
  if ((tid & 63) == 0) { dbg_[10 + (tid >> 6)] = wall_clock64(); dbg_[18 + (tid >> 6)] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); }
#endif
  __syncthreads();
  POCS_STEP_STAMP(6);
  if (tid < NC) store_wt(&a.partial[((size_t)r * a.slices + slot) * NC + tid], gmm_row_total<NC, RB>(sm.red[0], tid));
  drain_stores();
  __syncthreads();
  POCS_STEP_STAMP(7);
#if defined(POCS_STEP_STAMPS)
  if (tid == 0) { dbg_[8] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); dbg_[9] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); }
#endif
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(&a.ticket[(size_t)r * a.W + w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = (t == (unsigned)a.slices - 1u) ? 1 : 0;
    if (last) acquire_agent();
    sm.last = last;
  }
  __syncthreads();
  POCS_STEP_STAMP(3);
  if (__builtin_amdgcn_readfirstlane(sm.last) == 0) return;
  double* const l_mom = advance_ptrs(a, K, w + 1, r, sm.adv).l_mom;
  gmm_close_sums<K, RB>(a, w, r, sm.par[0], &sm.red[0][0][0], l_mom, tid, TB, [] { __syncthreads(); });
  POCS_STEP_STAMP(4);
  if (a.exchange_in_tail) {
    // sharded: the run's closer is also its messenger -- this shard's moments go to every rank, the world's
    // come back summed in rank order, and the mixture advances here; meanwhile the other engine's sampling
    // launch has the CUs this launch's finished blocks gave back (no launch, no host, between waypoints)
    __syncthreads();
    if (__hip_atomic_load(&a.sync[POCS_SYNC_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    if (!gmm_exchange_rows(a, a.xchg, K, w, r, l_mom, l_mom, tid, TB, &sm.last)) return;
  }
  if (a.advance_in_tail) advance_block(a, K, w + 1, r, sm.adv, sm.spec, true, tid, TB);     // starts with a barrier after staging
  __syncthreads();
  POCS_STEP_STAMP(5);
}

// ---------------------------------------------------------------------------------------------
// The whole run in ONE launch: every task (w, r, j) of the call's W waypoints x R runs x S slices,
// handed out in that order from a queue (one returning atomic per task and block).  A task of
// waypoint w > 0 needs `ready[r] >= w`, published by whoever closed (r, w-1) -- a task handed out
// earlier to a block that is running, so the wait always ends, however many blocks are resident.
//
// Nothing synchronises the grid, and nothing synchronises a block either: its waves run the block's
// tasks one after the other, each at its own pace, through a small pipeline kept in LDS.
//   * One wave is the LOADER of task n+1 (wave (n+1) mod 8: the role goes round): at the start of its
//     share of task n (when it has no sample stores in flight: a wave's loads, atomics and stores
//     complete in issue order, and write-through stores are slow to complete) it takes task n+1 from
//     the queue and, if that task's parameters are published, stages them and the culled obstacle
//     table into buffer (n+1) % 3; otherwise it tries again, and then waits, after its share of the
//     body.  seq[b] = n+2 hands the buffer to the other waves, who wait for it in LDS only.
//   * A wave that has sampled its share of task n and drained its stores counts itself in done[b].
//     The wave that counts last CLOSES the task for the block: rows -> the task's write-through
//     partial row -> drain -> ticket (r, w).  The wave that draws the run's last ticket acquires,
//     adds the run's partial rows, advances the mixture of run r to waypoint w+1 (one wave:
//     advance_wave) and publishes ready[r] = w+1 -- while its block's other waves and every other
//     block are already sampling.  freed[b] = n+1 gives the buffer back to wave 0.
// With R * S ~ 2.2 x the resident blocks per waypoint the tasks of waypoint w+1 come up in the queue
// when those of waypoint w have long been closed: the loader finds them published, and a block goes
// from one body straight into the next.  Every wait is bounded (2 s): a wave that gives up sets the
// call's give-up word and the block's quit word, everybody leaves at the next look, and the host
// reports POCS_E_DEVICE.
// ---------------------------------------------------------------------------------------------
// task buffers of k_gmm_run: three while two blocks of them fit a CU's 160 KB of LDS (K <= 5), two beyond
#define POCS_RUN_NB_OF(K) ((K) <= 5 ? 3 : 2)
__device__ __forceinline__ unsigned lds_load(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// LDS words written by another wave of the block: order our LDS accesses around them
__device__ __forceinline__ void lds_release() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); }
__device__ __forceinline__ void lds_acquire() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

// Spin (whole wave, uniform) until *word == want, in LDS.  false: the block is quitting.
__device__ __forceinline__ bool wait_lds(const unsigned* word, const unsigned want, unsigned* quit, unsigned* abort_word) {
  const unsigned long long t0 = wall_clock64();                         // 100 MHz
  unsigned polls = 0;
  while (lds_load(word) != want) {
    __builtin_amdgcn_s_sleep(1);
    if ((++polls & 63u) == 0u) {
      if (lds_load(quit) != 0u) return false;
      if (wall_clock64() - t0 > 200000000ull) {                         // 2 s
        __hip_atomic_store(abort_word, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_store(quit, 1u);
        return false;
      }
    }
  }
  lds_acquire();
  return true;
}
// ... until ready[r] >= need, in global memory (one lane polls; the caller acquires afterwards).
__device__ __forceinline__ bool wait_ready(unsigned* ready, const unsigned need, unsigned* quit, unsigned* abort_word) {
  const unsigned long long t0 = wall_clock64();
  unsigned polls = 0;
  while (__hip_atomic_load(ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(4);
    if ((++polls & 63u) == 0u) {
      if (lds_load(quit) != 0u) return false;
      if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
          wall_clock64() - t0 > 200000000ull) {
        __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_store(quit, 1u);
        return false;
      }
    }
  }
  return true;
}

// The loader wave of task `next` (= n + 1), all 64 lanes: stage it.  pending = the queue number
// already drawn for it (0xfffffffe: none yet).  blocking = wait for the buffer and for the task's
// parameters if need be.  Returns true when the task is staged and handed over (or the queue is
// exhausted: task_id >= total tells everybody to leave).
template <int K, int TB, int NB>
__device__ __forceinline__ bool gmm_stage_next(const pocs_gmm_launch& a, gmm_smem<K, TB, NB>& sm, const unsigned next,
                                               unsigned& pending, const bool blocking, const int lane) {
  const unsigned per_wp = (unsigned)a.nruns * (unsigned)a.slices;
  const unsigned total = per_wp * (unsigned)a.W;
  const int b = (int)(next % (unsigned)NB);
  if (next >= (unsigned)NB) {                        // the buffer's previous task (next - NB) must be closed
    const unsigned want = next - (unsigned)NB + 1u;
    if (lds_load(&sm.freed[b]) != want) {
#if defined(POCS_TASK_STAMPS)
      if (lane == 0) atomicAdd(&g_stamps[blocking ? 11 : 10], 1ull);
#endif
      if (!blocking) return false;
      if (!wait_lds(&sm.freed[b], want, &sm.quit, &a.sync[POCS_SYNC_ABORT])) return false;
    }
    lds_acquire();
  }
  if (pending == 0xfffffffeu) {
    unsigned t = 0u;
    if (lane == 0) {
      t = __hip_atomic_fetch_add(&a.sync[POCS_SYNC_HEAD], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__hip_atomic_load(&a.sync[POCS_SYNC_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) t = 0xffffffffu;
    }
    pending = (unsigned)__builtin_amdgcn_readfirstlane((int)t);
  }
  const unsigned t = pending;
  if (t < total) {
    const int w = __builtin_amdgcn_readfirstlane((int)(t / per_wp));
    const int r = __builtin_amdgcn_readfirstlane((int)((t - (unsigned)w * per_wp) / (unsigned)a.slices));
    if (w > 0) {
      unsigned have = 0u;
      if (lane == 0) have = __hip_atomic_load(&a.sync[POCS_SYNC_READY + r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((unsigned)__builtin_amdgcn_readfirstlane((int)have) < (unsigned)w) {
#if defined(POCS_TASK_STAMPS)
        if (lane == 0) atomicAdd(&g_stamps[blocking ? 15 : 14], 1ull);
#endif
        if (!blocking) return false;
        if (!wait_ready(&a.sync[POCS_SYNC_READY + r], (unsigned)w, &sm.quit, &a.sync[POCS_SYNC_ABORT])) return false;
      }
      acquire_agent();                               // the poll matched: ONE acquire, then the loads
    }
    const int slot = (int)(t - (unsigned)w * per_wp) - r * a.slices;
    stage_batched<2>(sm.par[b], K * POCS_PARAM_STRIDE, lane, 64, [&](int j) -> double {
      return load_wt(&a.param[((size_t)r * a.W + w) * (K * POCS_PARAM_STRIDE) + j]); });
    if (lane == 0) {
      long long cb, ce;
      gmm_slice_chunks(a, slot, &cb, &ce);
      sm.seed[b] = a.hdr[r].seed;
      sm.c_begin[b] = cb; sm.c_end[b] = ce;
      sm.tw[b] = w; sm.tr[b] = r; sm.tslot[b] = slot;
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
    gmm_cull(a, sm, b, lane);
  }
  if (lane == 0) {
    sm.done[b] = 0u;
    sm.task_id[b] = t;
  }
  lds_release();
  if (lane == 0) lds_store(&sm.seq[b], next + 1u);
  pending = 0xfffffffeu;
  return true;
}

template <int K, bool STORE, int TB>
__global__ __launch_bounds__(TB, POCS_GMM_BLOCKS_PER_CU * TB / 256) void k_gmm_run(pocs_gmm_launch a) {
  constexpr int NB = POCS_RUN_NB_OF(K);
  typedef gmm_smem<K, TB, NB> smem_t;
  constexpr int NC = smem_t::NC, RB = smem_t::RB, NW = TB / 64;
  __shared__ smem_t sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned per_wp = (unsigned)a.nruns * (unsigned)a.slices;
  const unsigned total = per_wp * (unsigned)a.W;
  gmm_stage_static(a, sm);
  if (tid < NB) { sm.seq[tid] = 0u; sm.done[tid] = 0u; sm.freed[tid] = 0u; sm.task_id[tid] = 0u; }
  if (tid == 0) { sm.lock = 0u; sm.quit = 0u; }
  __syncthreads();                                   // the only barrier of the launch
  unsigned pending = 0xfffffffeu;
  if (wave == 0 && !gmm_stage_next(a, sm, 0u, pending, true, lane)) return;
#if defined(POCS_TASK_STAMPS)
  unsigned long long st_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = wall_clock64();
  const unsigned long long t_begin_ = last_;
#endif
  for (unsigned n = 0;; ++n) {
    const int b = (int)(n % (unsigned)NB);
    // the loader of task n + 1 is wave (n + 1) mod NW: the role goes round, so that its round trips (and
    // the closing of a task, which falls to whichever wave finishes last) delay every wave alike
    const bool loader = wave == (int)((n + 1u) % (unsigned)NW);
    if (!wait_lds(&sm.seq[b], n + 1u, &sm.quit, &a.sync[POCS_SYNC_ABORT])) break;
    POCS_STAMP(0);
    const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)sm.task_id[b]);
    if (t >= total) break;
    const int w = __builtin_amdgcn_readfirstlane(sm.tw[b]), r = __builtin_amdgcn_readfirstlane(sm.tr[b]);
    const int slot = __builtin_amdgcn_readfirstlane(sm.tslot[b]);
    const uint64_t seed = (uint64_t)uniform64((long long)sm.seed[b]);
    const long long c_begin = uniform64(sm.c_begin[b]), c_end = uniform64(sm.c_end[b]);
    bool staged = true;
    if (loader) staged = gmm_stage_next(a, sm, n + 1u, pending, false, lane);
    POCS_STAMP(1);
#if defined(POCS_TASK_STAMPS)
    gmm_body<K, STORE, true, TB>(a, &sm.tab, sm.par[b], sm.keep[b], __builtin_amdgcn_readfirstlane(sm.nkeep[b]), sm.red[b], w, r,
                                 seed, c_begin, c_end, st_, last_);
#else
    gmm_body<K, STORE, true, TB>(a, &sm.tab, sm.par[b], sm.keep[b], __builtin_amdgcn_readfirstlane(sm.nkeep[b]), sm.red[b], w, r,
                                 seed, c_begin, c_end);
#endif
    // this wave's share is sampled: drain its stores, count it in
    drain_stores();
    POCS_STAMP(5);
    lds_release();
    unsigned pos = 0u;
    if (lane == 0) pos = __hip_atomic_fetch_add(&sm.done[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)pos) == (unsigned)NW - 1u) {
    // ---- the block's last wave for this task closes it
    lds_acquire();
    for (int c = lane; c < NC; c += 64)
      store_wt(&a.partial[((size_t)r * a.slices + slot) * NC + c], gmm_row_total<NC, RB>(sm.red[b], c));
    drain_stores();
    unsigned tk = 0u;
    if (lane == 0) tk = __hip_atomic_fetch_add(&a.ticket[(size_t)r * a.W + w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)tk) == (unsigned)a.slices - 1u) {
      // ---- ... and the run's last block for this waypoint closes (r, w)
#if defined(POCS_TASK_STAMPS)
      const unsigned long long f0_ = wall_clock64();
#endif
      acquire_agent();
      bool mine = true;                              // the advance scratch is one wave's at a time
      {
        const unsigned long long t0 = wall_clock64();
        for (;;) {
          unsigned old = 1u;
          if (lane == 0) { unsigned exp = 0u; old = __hip_atomic_compare_exchange_strong(&sm.lock, &exp, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 0u : 1u; }
          if (__builtin_amdgcn_readfirstlane((int)old) == 0) break;
          __builtin_amdgcn_s_sleep(2);
          if (lds_load(&sm.quit) != 0u || wall_clock64() - t0 > 200000000ull) { mine = false; break; }
        }
      }
      if (!mine) {
        __hip_atomic_store(&a.sync[POCS_SYNC_ABORT], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_store(&sm.quit, 1u);
        break;
      }
      lds_acquire();
#if defined(POCS_TASK_STAMPS)
      const unsigned long long f1_ = wall_clock64();
#endif
      double* const l_mom = advance_ptrs(a, K, w + 1, r, sm.adv).l_mom;
      gmm_close_sums<K, RB>(a, w, r, sm.par[b], &sm.red[b][0][0], l_mom, lane, 64,
                            [] { __threadfence_block(); __builtin_amdgcn_wave_barrier(); });
#if defined(POCS_TASK_STAMPS)
      const unsigned long long f2_ = wall_clock64();
#endif
      if (w + 1 < a.W) {
        advance_wave(a, K, w + 1, r, sm.adv, lane);                    // ends with its stores drained
        if (lane == 0)
          __hip_atomic_store(&a.sync[POCS_SYNC_READY + r], (unsigned)(w + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#if defined(POCS_TASK_STAMPS)
      if (lane == 0) {
        const unsigned long long f3_ = wall_clock64();
        atomicAdd(&g_stamps[16], f1_ - f0_); atomicAdd(&g_stamps[17], f2_ - f1_); atomicAdd(&g_stamps[18], f3_ - f2_);
        atomicAdd(&g_stamps[19], 1ull);
      }
#endif
      lds_release();
      if (lane == 0) lds_store(&sm.lock, 0u);
    }
    lds_release();
    if (lane == 0) lds_store(&sm.freed[b], n + 1u);
    }
    POCS_STAMP(6);
    // the loader could not stage the next task before its share of this one (not published yet, or the
    // buffer still in use): now it waits -- AFTER counting itself in above, the closing of this very
    // task may be what the next one is waiting for
    if (loader && !staged && !gmm_stage_next(a, sm, n + 1u, pending, true, lane)) break;
    POCS_STAMP(7);
#if defined(POCS_TASK_STAMPS)
    st_[9] += 1;
#endif
  }
#if defined(POCS_TASK_STAMPS)
  if (lane == 0) {
    const unsigned long long life_ = wall_clock64() - t_begin_;
    for (int i = 0; i < 10; ++i) atomicAdd(&g_stamps[i], st_[i]);
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
  unsigned long long z[32] = {0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z);
  const hipError_t e = launch_gmm_run_plain(K, nblk, a, s);
  (void)hipStreamSynchronize(s);
  unsigned long long h[32];
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof h) == hipSuccess && h[13] > 0) {
    static const char* names[8] = {"wait-seq", "decode+stage-try", "prologue", "loop", "flush", "drain", "count+close", "stage-blocking"};
    const double nw = (double)h[13], tasks = (double)h[9];
    fprintf(stderr, "stamps: %g waves, %g wave-tasks, mean wave lifetime %.1f us; per wave-task (us):", nw, tasks, 0.01 * h[12] / nw);
    for (int i = 0; i < 8; ++i) fprintf(stderr, " %s %.2f", names[i], 0.01 * (double)h[i] / tasks);
    fprintf(stderr, "; of %g tasks the loader found: buffer busy %llu (again when blocking %llu), not published %llu (again %llu)\n",
            tasks / 8.0, h[10], h[11], h[14], h[15]);
    if (h[19]) fprintf(stderr, "stamps: %llu run-waypoints closed; per closing (us): acquire+lock %.2f, adding the rows %.2f, advance+publish %.2f\n", h[19],
                       0.01 * (double)h[16] / (double)h[19], 0.01 * (double)h[17] / (double)h[19], 0.01 * (double)h[18] / (double)h[19]);
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
// Plain streaming FILL: what a write-only stream reaches on this GPU -- the GMM kernels read nothing, so
// this, not the copy rate, is the bandwidth ceiling they could run into (bench.py "fill_GBps").
__global__ __launch_bounds__(POCS_BLOCK) void k_fill(double2* __restrict__ dst, long long n, double v) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  v2d* d = reinterpret_cast<v2d*>(dst);
  const v2d x = {v, v};
  const long long stride = (long long)gridDim.x * POCS_BLOCK * 4;
  for (long long i = (long long)blockIdx.x * POCS_BLOCK * 4 + threadIdx.x; i < n; i += stride) {
#pragma unroll
    for (int u = 0; u < 4; ++u) if (i + u * POCS_BLOCK < n) __builtin_nontemporal_store(x, d + i + u * POCS_BLOCK);
  }
}
hipError_t pocs_launch_fill(void* dst, long long bytes, hipStream_t s) {
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(POCS_BLOCK), 0, s, (double2*)dst, bytes / 16, 1.5);
  return hipGetLastError();
}
hipError_t pocs_launch_copy(const void* src, void* dst, long long bytes, hipStream_t s) {
  hipLaunchKernelGGL(k_copy, dim3(8192), dim3(POCS_BLOCK), 0, s, (const double2*)src, (double2*)dst, bytes / 16);
  return hipGetLastError();
}

hipError_t pocs_launch_gmm_exchange(int K, const pocs_gmm_launch& a, const pocs_xchg_dev& x, hipStream_t s) {
  hipLaunchKernelGGL(k_gmm_exchange, dim3(a.nruns), dim3(128), 0, s, a, x, K);
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
