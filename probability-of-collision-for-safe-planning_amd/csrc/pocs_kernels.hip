// pocs_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the hot path.
//
//   k_gmm_step      S1+C1+T1  one waypoint of truncateGMM (MCSimulator.h:570-642) in ONE launch, for
//                             every run of a batch of independent estimations: the launch's units
//                             (run, virtual slice) dealt evenly to the blocks.  A block:
//                             head  log/sector tables, obstacle table and this waypoint's sampler
//                                   parameters -> LDS; exact culling of the obstacle table against
//                                   the mixture's bounding box, the kept records' broad phase
//                                   tightened to the run's range of headings;
//                             body  GM_Model::sampleNPoints (GM_Model.h:83-116) + checkMatrixCollisions
//                                   (:241-253) + the moment sums (:592-611), fused, one PAIR of
//                                   samples per thread-iteration: a sample is born, tested and folded
//                                   into its component's (n, sum x, sum x x^T) in registers; pose
//                                   and flag are streamed out once (24 B + 2 B).  A wave whose 128
//                                   samples lie in one component block (nearly always) runs the
//                                   iteration's scalar-component form in an inner loop of its own;
//                                   at the end of every virtual slice the wave's lane chains become a
//                                   wave sum in LDS (transposition, no barrier);
//                             tail  wave sums -> write-through rows per (run, virtual slice) -> ticket ->
//                                   the last block of a run to arrive adds the run's rows in a fixed
//                                   order and advances the mixture to the next waypoint: truncated
//                                   mean/cov, weights (:597-629), per-component EKF predict/update
//                                   (:766-771, :804-812), Cholesky -- on one GPU right away, sharded
//                                   after it has exchanged the run's moments with the other ranks (IPC
//                                   slots, one hop over xGMI) in the same tail.
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
#include <stdio.h>

namespace {

// Row sums by DPP: four steps (pairs, quads, half rows, rows) leave every lane of a 16-lane row
// holding its row's sum.  Every lane has a valid source in all four patterns, so `old` is never
// used; the shape is fixed, hence bitwise reproducible run to run.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  // (`old` = 0 with bound_ctrl: every lane has a valid source in the patterns used here, so `old` is never
  // taken, and the compiler need not copy the source to protect it)
  const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
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

// Stage the log / sector tables (12 KB) into LDS.  The log table's 1/c entries are DOUBLED on the way in:
// the device form of pocs_radius2_unit32 multiplies them with the mantissa in [1/2, 1) (pocs_math.h).
__device__ __forceinline__ void stage_tables(const pocs_tables* __restrict__ g, pocs_tables* s_tab) {
  const double* src = reinterpret_cast<const double*>(g);
  double* dst = reinterpret_cast<double*>(s_tab);
  constexpr int NLG = (int)(sizeof(g->lg) / sizeof(double));
  for (int j = threadIdx.x; j < (int)(sizeof(pocs_tables) / sizeof(double)); j += blockDim.x)
    dst[j] = (j < NLG && (j & 1) == 0) ? 2.0 * src[j] : src[j];
}

// The same in two steps for a block of TB threads (k_gmm_step's heads): the loads, into registers -- and, once everything
// else the head needs has been requested behind them, the stores.  (stage_tables' loop has a run-time stride: the
// compiler keeps it a loop and waits for every load before it issues the next, three memory round trips one after the
// other for 24 bytes per thread.)
constexpr int POCS_TABLE_DOUBLES = (int)(sizeof(pocs_tables) / sizeof(double));
template <int TB> struct table_regs { static constexpr int N = (POCS_TABLE_DOUBLES + TB - 1) / TB; };
template <int TB>
__device__ __forceinline__ void request_tables(const pocs_tables* __restrict__ g, const int tid, double (&v)[table_regs<TB>::N]) {
  const double* src = reinterpret_cast<const double*>(g);
#pragma unroll
  for (int u = 0; u < table_regs<TB>::N; ++u) { const int j = tid + u * TB; v[u] = (j < POCS_TABLE_DOUBLES) ? src[j] : 0.0; }
}
template <int TB>
__device__ __forceinline__ void commit_tables(pocs_tables* s_tab, const int tid, const double (&v)[table_regs<TB>::N]) {
  double* dst = reinterpret_cast<double*>(s_tab);
  constexpr int NLG = (int)(sizeof(s_tab->lg) / sizeof(double));
#pragma unroll
  for (int u = 0; u < table_regs<TB>::N; ++u) {
    const int j = tid + u * TB;
    if (j < POCS_TABLE_DOUBLES) dst[j] = (j < NLG && (j & 1) == 0) ? 2.0 * v[u] : v[u];
  }
}
// everything requested so far is in flight before anything that follows is issued (loads do not sink below it, stores do
// not rise above it): the head's requests, then ONE wait
__device__ __forceinline__ void requests_issued() { asm volatile("" ::: "memory"); }

// The head of an MC block (POCS_BLOCK threads): the collision world (obstacle records, footprint, M) and the 4 KB sector table --
// the MC kernels only evaluate the footprint heading -- into LDS.  Every load is issued before the first is waited for: written
// as copy loops with the block size as their stride, the compiler kept them loops of load - wait - store, four dependent
// memory round trips in front of every block's first particle.  (The obstacle array has its full size whatever M is: all of
// it is requested, M need not be known first.)
__device__ __forceinline__ void stage_mc_head(const pocs_env_dev* __restrict__ env, const pocs_tables* __restrict__ g, double* s_obs,
                                              pocs_footprint* s_fp, int* s_M, pocs_tables* s_tab) {
  constexpr int NO = POCS_MAX_OBSTACLES * POCS_OBS_STRIDE, NS = (int)(sizeof(g->sc) / sizeof(double));
  constexpr int UO = (NO + POCS_BLOCK - 1) / POCS_BLOCK, US = (NS + POCS_BLOCK - 1) / POCS_BLOCK;
  const int tid = threadIdx.x;
  const double* src = &g->sc[0][0];
  double vo[UO], vs[US];
#pragma unroll
  for (int u = 0; u < UO; ++u) { const int i = tid + u * POCS_BLOCK; vo[u] = i < NO ? env->obs[i] : 0.0; }
#pragma unroll
  for (int u = 0; u < US; ++u) { const int i = tid + u * POCS_BLOCK; vs[u] = i < NS ? src[i] : 0.0; }
  const pocs_footprint fp = env->fp;
  const int M = env->M;
  asm volatile("" ::: "memory");                    // (requests_issued, defined further down)
  double* dst = &s_tab->sc[0][0];
#pragma unroll
  for (int u = 0; u < UO; ++u) { const int i = tid + u * POCS_BLOCK; if (i < NO) s_obs[i] = vo[u]; }
#pragma unroll
  for (int u = 0; u < US; ++u) { const int i = tid + u * POCS_BLOCK; if (i < NS) dst[i] = vs[u]; }
  if (tid == 0) { *s_fp = fp; *s_M = M; }
}

// ---------------------------------------------------------------------------------------------
// Hand-offs between workgroups (the rows of a run's virtual slices -> the last arriver, inside a launch; mixture
// state and sampler parameters -> the blocks of the next waypoint's launch).  cdna_hip_programming.md Guideline 16,
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
typedef double v2d __attribute__((ext_vector_type(2)));
// The sample stream: SGPR base + 32-bit lane offset, non-temporal (the compiler, left to itself,
// builds a 64-bit address per lane and store: four vector adds per iteration)
__device__ __forceinline__ void store16_nt(const void* base_uniform, unsigned lane_bytes, v2d v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_bytes), "v"(v), "s"(base_uniform) : "memory");
}
__device__ __forceinline__ void store4_nt(const void* base_uniform, unsigned lane_bytes, int v) {
  asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(lane_bytes), "v"(v), "s"(base_uniform) : "memory");
}
// lane `l`'s value of v, in every lane
__device__ __forceinline__ double lane_value(double v, int l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
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

// Diagnostic hooks (phase stamps, timing-only ablations of the sampling body): real only in a -DPOCS_TUNING build
// (csrc/pocs_tuning.h, tools/ablate.sh); the shipped library sees the no-ops / pass-throughs below.
#ifdef POCS_TUNING
#include "pocs_tuning.h"
#else
#define POCS_STAMP_BEGIN() do { } while (0)
#define POCS_STAMP(i) do { } while (0)
#define POCS_STAMP_COUNT(i) do { } while (0)
#define POCS_ADV_STAMP_BEGIN() do { } while (0)
#define POCS_ADV_STAMP(i) do { } while (0)
#define POCS_TUNE_NORMALS(...) __VA_ARGS__
#define POCS_TUNE_COLLIDE(...) __VA_ARGS__
#define POCS_TUNE_COLLIDE_STATS() do { } while (0)
#define POCS_TUNE_SKIP_MOMENTS false
#define POCS_TUNE_MOMENTS_ALT() do { } while (0)
#endif

// LDS scratch of the mixture advance (doubles): state[w-1], moments, chain record, sensor, state[w], param[w];
// and of the speculated component counts.
#define POCS_ADV_SCRATCH(K) ((K) * (2 * POCS_STATE_STRIDE + POCS_NMOM + POCS_PARAM_STRIDE) + POCS_CHAIN_STRIDE + \
                             (int)(sizeof(pocs_sensor) / sizeof(double)))
#define POCS_SPEC_SCRATCH(K) ((K) * (POCS_STATE_STRIDE + 2))

// LDS of k_gmm_step.  A block works through a contiguous range of the launch's UNITS -- (run, virtual slice)
// pairs, pocs_kernels.h -- that may cross from one run into the next: everything per run is held twice.
//   tr     the wave's transpose scratch of flush_unit (5 rows of 64 lane values, pitch 66); before the first
//          flush the same bytes hold the full obstacle table the culling reads, after the last one the
//          closer's staging rows (gmm_close_sums)
//   slot   wave sums (survivors, nine sums) of the unit's FIRST component, per virtual slice held and wave;
//          once the rows are out, the mixture advance's scratch
//   xtra   ... of a component that STARTS inside the unit (a wave meets the start of a component once per run)
template <int K, int TB>
struct gmm_smem {
  static constexpr int NC = K * POCS_NMOM;
  static constexpr int NW = TB / 64;
  static constexpr int SUB = POCS_GMM_SUB;
  alignas(16) pocs_tables tab;                                       // 12 KB log / sector tables, staged once per block
  alignas(16) double keep[2][POCS_MAX_OBSTACLES * POCS_OBS_STRIDE];  // obstacle table culled for the block's (up to) two runs
  alignas(16) double par[2][K * POCS_PARAM_STRIDE];                  // sampler parameters of (run, waypoint)
  alignas(16) double tr[NW][POCS_FLUSH_ROWS][POCS_FLUSH_PITCH];
  alignas(16) double slot[SUB][NW][POCS_UNIT_SUMS];
  double xtra[2][NW][K][POCS_UNIT_SUMS];
  int kf[SUB][NW];                                                   // the component slot[..] belongs to
  int xj[2][NW][K];                                                  // the held virtual slice xtra[..] belongs to (-1: none)
  unsigned long long seed[2];                                        // the two runs' seeds
  int nkeep[2];
  int last[2];                                                       // this block drew the last ticket of its run 0 / 1
  static_assert(sizeof(double) * NW * POCS_FLUSH_ROWS * POCS_FLUSH_PITCH >= sizeof(double) * POCS_MAX_OBSTACLES * POCS_OBS_STRIDE,
                "the obstacle table is staged in the transpose scratch");
  static_assert(NW * POCS_FLUSH_ROWS * POCS_FLUSH_PITCH >= 16 * NC, "the closer's staging rows live in the transpose scratch");
  static_assert(SUB * NW * POCS_UNIT_SUMS >= POCS_ADV_SCRATCH(K) + POCS_SPEC_SCRATCH(K), "the advance's scratch lives in the slots");
  __device__ __forceinline__ double* obs() { return &tr[0][0][0]; }
  __device__ __forceinline__ double* stage() { return &tr[0][0][0]; }
  __device__ __forceinline__ double* adv() { return &slot[0][0][0]; }
  __device__ __forceinline__ double* spec() { return &slot[0][0][0] + POCS_ADV_SCRATCH(K); }
};

// Mixture bookkeeping of waypoint `w` (pocs_gmm_advance_component / pocs_gmm_normalise): every input
// (state[w-1], the reduced moments of w-1, the chain record of step w-1, the sensor) is first brought
// to LDS in ONE round trip, lanes < K of one wave then take one component each (truncated mean /
// covariance, EKF predict + update, Cholesky); lane 0 normalises, draws the component counts, and the
// wave writes state[w] / param[w] back write-through.  Run by a whole block (k_gmm_advance,
// k_gmm_step: a lane of a second wave draws the counts meanwhile, on the premise -- checked -- that
// no factorisation fails).
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
// (request / commit: the batch of four loads per thread that starts at index i0, and their stores -- a caller with other
// requests to make puts them between the two, so that all of them share one round trip)
__device__ __forceinline__ int advance_stage_count(const adv_ptrs& p, const bool load_mom) {
  constexpr int SEN = (int)(sizeof(pocs_sensor) / sizeof(double));
  return p.ss + POCS_CHAIN_STRIDE + SEN + (load_mom ? p.NC : 0);
}
//   ONE_LOAD: one load per element, its address chosen by the element's range, L1-bypassing for all four sources (state[w-1]
//   needs it, the others do not mind) -- for the closer, where around four different loads in an if / else-if chain the compiler
//   put a wait behind each branch (three dependent round trips); the lone form's heads keep the chain (no waits there, and the
//   chain record and the sensor come through the caches: measured 0.17 us per waypoint)
template <bool ONE_LOAD = false>
__device__ __forceinline__ void advance_request(const adv_ptrs& p, const bool load_mom, const int i0, const int nthreads, double (&v)[4],
                                                const bool wanted = true) {
  constexpr int SEN = (int)(sizeof(pocs_sensor) / sizeof(double));
  const int ss = p.ss, n = wanted ? advance_stage_count(p, load_mom) : 0;     // (not wanted: no lane loads anything -- and no branch round the loads)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = i0 + u * nthreads;
    if (ONE_LOAD) {
      const double* src = i < ss ? &p.g_prev[i]
                        : i < ss + POCS_CHAIN_STRIDE ? &p.g_ch[i - ss]
                        : i < ss + POCS_CHAIN_STRIDE + SEN ? &p.g_sen[i - ss - POCS_CHAIN_STRIDE]
                        : &p.g_mom[i - ss - POCS_CHAIN_STRIDE - SEN];
      v[u] = i < n ? load_wt(src) : 0.0;
    } else {
      v[u] = 0.0;
      if (i < ss) v[u] = load_wt(&p.g_prev[i]);
      else if (i < ss + POCS_CHAIN_STRIDE) v[u] = p.g_ch[i - ss];
      else if (i < ss + POCS_CHAIN_STRIDE + SEN) v[u] = p.g_sen[i - ss - POCS_CHAIN_STRIDE];
      else if (i < n) v[u] = p.g_mom[i - ss - POCS_CHAIN_STRIDE - SEN];
    }
  }
}
__device__ __forceinline__ void advance_commit(const adv_ptrs& p, const bool load_mom, const int i0, const int nthreads, const double (&v)[4]) {
  constexpr int SEN = (int)(sizeof(pocs_sensor) / sizeof(double));
  const int ss = p.ss, n = advance_stage_count(p, load_mom);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = i0 + u * nthreads;
    if (i < ss) p.l_prev[i] = v[u];
    else if (i < ss + POCS_CHAIN_STRIDE + SEN) p.l_ch[i - ss] = v[u];              // l_ch | l_sen are contiguous
    else if (i < n) p.l_mom[i - ss - POCS_CHAIN_STRIDE - SEN] = v[u];
  }
}
__device__ __forceinline__ void advance_stage(const pocs_gmm_launch& a, int K, int w, int r, double* scratch,
                                              bool mom_in_lds, int tid, int nthreads) {
  const adv_ptrs p = advance_ptrs(a, K, w, r, scratch);
  const bool load_mom = w > 0 && !mom_in_lds;
  // index space: [0, ss) state | [ss, ss + CH + SEN) chain record, sensor | then (only if wanted) the moments;
  // l_mom is NOT touched when the caller has put the moments there
  const int n = advance_stage_count(p, load_mom);
  for (int i0 = tid; i0 < n; i0 += nthreads * 4) {
    double v[4];
    advance_request(p, load_mom, i0, nthreads, v);
    requests_issued();
    advance_commit(p, load_mom, i0, nthreads, v);
  }
}
// the largest index space of advance_stage: a block of TB threads with 4 TB >= this stages it in ONE batch
#define POCS_ADV_STAGE_MAX (POCS_MAX_GAUSSIANS * (POCS_STATE_STRIDE + POCS_NMOM) + POCS_CHAIN_STRIDE + (int)(sizeof(pocs_sensor) / sizeof(double)))

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
// if there are any and their premise held), write-through stores of state[w] / param[w].
__device__ __forceinline__ void advance_finish(const pocs_gmm_launch& a, int K, int w, int r, int lane, double* scratch,
                                               const double* spec, const bool publish = true) {
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
  if (!publish) return;                              // (lone call: one block of the launch writes the records out)
  for (int j = lane; j < p.ss; j += 64) store_wt(&p.g_state[(size_t)w * p.ss + j], p.l_next[j]);
  for (int j = lane; j < p.ps; j += 64) store_wt(&p.g_param[(size_t)w * p.ps + j], p.l_par[j]);
  // (not drained: nothing inside this launch reads the records -- state[w] / param[w] are for the NEXT waypoint's launch, behind
  // the kernel boundary; waiting for the write-through stores here kept every closer 1.2 us longer in the launch's tail)
}

// The whole advance to waypoint w by a block of >= 128 threads (every thread calls it).
// staged: the caller has issued advance_stage already (and a barrier since); publish: state[w] / param[w] go out
// to global memory as well (always, except for all but one block of a lone call's launch).
struct advance_no_side_job { __device__ __forceinline__ void operator()() const {} };
// `side_job`: run by the threads tid >= 128 -- the waves that otherwise wait at the barrier below for the components'
// serial chain (wave 0) and the count lane (wave 1) -- with nothing of the advance's scratch in it (the lone form's
// heads draw the first iterations' normals there, k_gmm_step)
// `post_job`: run by wave 1 (64 <= tid < 128; its lane 0 has drawn the counts by then) beside advance_finish, with nothing in it
// that the normalisation still changes -- means and Cholesky factors of param[w] are final once the components are through
// (the lone form's heads cull the obstacle table there)
template <typename SideJob = advance_no_side_job, typename PostJob = advance_no_side_job>
__device__ __forceinline__ void advance_block(const pocs_gmm_launch& a, int K, int w, int r, double* adv, double* spec,
                                              bool mom_in_lds, int tid, int nthreads, const bool staged = false,
                                              const bool publish = true, SideJob side_job = SideJob(), PostJob post_job = PostJob()) {
  POCS_ADV_STAMP_BEGIN();
  if (!staged) {
    advance_stage(a, K, w, r, adv, mom_in_lds, tid, nthreads);
    __syncthreads();
  }
  POCS_ADV_STAMP(8);
  if (tid < 64) advance_components(a, K, w, r, tid, adv);
  else if (tid == 64 && w > 0) speculate_counts(a, K, w, r, adv, spec);
  else if (tid >= 128) side_job();
  POCS_ADV_STAMP(9);
  __syncthreads();
  POCS_ADV_STAMP(10);
  if (tid < 64) advance_finish(a, K, w, r, tid, adv, w > 0 ? spec : nullptr, publish);
  else if (tid < 128) post_job();
  POCS_ADV_STAMP(11);
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
// with the parity of a running count of exchanges that is the same on every rank (calls x W + waypoint,
// pocs_xchg_dev::parity -- NOT the waypoint's own parity: with an odd W the last exchange of a call and
// the first of the next would share a slot): a rank can only be one exchange ahead of another.  The wait is bounded
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
__device__ __forceinline__ bool gmm_exchange_rows(const pocs_gmm_launch& a, const pocs_xchg_dev& x, const unsigned long long epoch, const int parity,
                                                  const int K, const int w, const int r,
                                                  const double* mine, double* l_mom, const int tid, const int nthreads, int* s_ok) {
  const int NC = K * POCS_NMOM;
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
    __hip_atomic_store(xchg_flag(x.buf[tid], parity, x.rank, r), epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // every rank's row of this waypoint has landed in MY buffer?
  if (tid == 0) *s_ok = 1;
  __syncthreads();
  if (tid < x.world) {
    const unsigned long long* f = xchg_flag(x.buf[x.rank], parity, tid, r);
    const unsigned long long t0 = wall_clock64();
    unsigned polls = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
      __builtin_amdgcn_s_sleep(8);
      if ((++polls & 255u) == 0u && wall_clock64() - t0 > 3000000000ull) {      // 30 s: ranks of a cold node start seconds apart
        __hip_atomic_store(&a.sync[POCS_SYNC_ABORT], 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_ok = 0;
        break;
      }
    }
    // how long this closer waited for rank `tid`'s row (10 ns ticks; its own row: no time): the longest of them is
    // what the exchange cost this (run, waypoint) -- pocs_get_exchange_wait, for a scaling run that explains itself
    const unsigned long long dt = wall_clock64() - t0;
    __hip_atomic_fetch_max(&a.xwait[(size_t)r * a.W + w], (unsigned)(dt < 0xffffffffull ? dt : 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < 64) {                                          // ONE wave, the one that polled: drop this XCD's stale lines
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // system scope
    drain_stores();                                        // the invalidate completes before the barrier releases the readers
  }
  __syncthreads();
  if (!*s_ok) return false;
  // the slots of my buffer, added in rank order -- every rank's value REQUESTED before the first is waited for (as a loop over
  // the world with the addition in it, the compiler waits for each system-scope load before it issues the next: eight
  // dependent round trips per waypoint on eight GPUs)
  for (int c = tid; c < NC; c += nthreads) {
    double v[POCS_XCHG_MAX_WORLD];
#pragma unroll
    for (int q = 0; q < POCS_XCHG_MAX_WORLD; ++q)
      v[q] = q < x.world ? __longlong_as_double((long long)__hip_atomic_load(
                 reinterpret_cast<const unsigned long long*>(xchg_row(x.buf[x.rank], parity, q, r) + c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM))
                         : 0.0;
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    static_assert(POCS_XCHG_MAX_WORLD == 8, "eight values pinned");
    double tot = 0.0;
#pragma unroll
    for (int q = 0; q < POCS_XCHG_MAX_WORLD; ++q) if (q < x.world) tot += v[q];
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
  if (!gmm_exchange_rows(a, x, x.epoch, x.parity, K, w, r, a.moments + ((size_t)w * a.nruns + r) * NC, l_mom, tid, 128, &s_ok)) return;
  if (w + 1 < a.W) advance_block(a, K, w + 1, r, s_adv, s_spec, true, tid, 128);     // starts with a barrier after staging
}

// ---------------------------------------------------------------------------------------------
// The moment sums have ONE fixed shape, whatever the launch looks like ("summation
// tree"; oracle/pocs_oracle.c restates it and the two agree bit for bit):
//   lane chain   a lane's samples of one component inside one UNIT-WAVE -- wave v (tid / 64) of virtual slice
//                j of the run, over the slice's chunks in order, sample 2 lp before 2 lp + 1 -- accumulated
//                sequentially: sums += x, fma(x, x, sum) ...; survivors counted as integers;
//   wave sum     the 64 lane chains: eight runs of eight lanes added in lane order, then
//                ((g0 + g1) + (g2 + g3)) + ((g4 + g5) + (g6 + g7))                         (flush_unit)
//   row          of (virtual slice, component): the eight wave sums in wave order          (gmm_emit_rows)
//   total        the run's VS rows as sixteen interleaved partial sums, then those in order (gmm_close_sums)
// A run always has the same VS virtual slices (a function of the shard's sample count only), so the
// result does not depend on how many runs share a launch, on the blocks a launch uses, or on which
// block or wave worked on which slice: a batch of R runs, run-ahead and R single calls give the same bits.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double oct_sum(double s) {     // lanes 8 j .. 8 j + 7 hold g0 .. g7 -> all hold the sum above
  s += dpp_f64<0xB1>(s);     // quad_perm [1,0,3,2]
  s += dpp_f64<0x4E>(s);     // quad_perm [2,3,0,1]
  s += dpp_f64<0x141>(s);    // row_half_mirror
  return s;
}

// A wave leaves component `k` of the unit it is working on (held virtual slice `tl`, run buffer `rb`): its
// lane chains -> the wave sum -> LDS (slot[tl][wave] if this is the unit's first component, otherwise the
// wave's xtra entry of k), and the chains restart.  By LDS transposition, five then four sums at a
// time: every lane writes its values, lane 8 j + q adds lanes 8 q .. 8 q + 7 of sum j, oct_sum adds the
// eight q.  One wave: the LDS executes a wave's instructions in order, nothing else synchronises.
template <int K, int TB>
__device__ __forceinline__ void flush_unit(gmm_smem<K, TB>& sm, const int wave, const int lane, const int rb, const int tl,
                                           const int k, bool& first, double (&acc)[9], int& nfree) {
  double* const T = &sm.tr[wave][0][0];
  double* const dst = first ? &sm.slot[tl][wave][0] : &sm.xtra[rb][wave][k][0];
  const int j = lane >> 3, q = lane & 7;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int nv = pass == 0 ? 5 : 4, v0 = pass == 0 ? 0 : 5;
#pragma unroll
    for (int i = 0; i < nv; ++i) T[i * POCS_FLUSH_PITCH + lane] = acc[v0 + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double s = 0.0;
    if (j < nv) {
      const double* p = &T[j * POCS_FLUSH_PITCH + 8 * q];
      s = p[0]; s += p[1]; s += p[2]; s += p[3]; s += p[4]; s += p[5]; s += p[6]; s += p[7];
    }
    s = oct_sum(s);
    if (j < nv && q == 0) dst[1 + v0 + j] = s;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    dst[0] = (double)nfree;
    if (first) sm.kf[tl][wave] = k; else sm.xj[rb][wave][k] = tl;
  }
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = 0.0;
  nfree = 0;
  first = false;
}

// Cull the obstacle table against the bounding box of the mixture staged in par[rb] (ONE wave, all 64
// lanes).  A Box-Muller normal is bounded: u >= 2^-32 gives |z| <= sqrt(64 ln 2) < 6.661
// (pocs_normal_pair_w2; 6.67 leaves 0.1 % for the rounding of radius * cos), so every pose the run can
// draw lies within mean_k +- 6.67 (|L00|, |L10|+|L11|) of some component; an obstacle whose inflated
// box (the broad phase of pocs_box_hit) misses that region is rejected by the broad phase for every
// sample, so dropping it here changes no flag.
//
// The same bound on the heading makes the broad phase of the kept records tighter than the table's: the
// table inflates an obstacle's box by the footprint's bounding RADIUS (any heading); a run whose
// headings all lie in [t_lo, t_hi] needs only the footprint's largest half-extent along world x and
// along world y over that range (two convex sets that touch overlap in every projection).  Where the
// robot's heading is known to a fraction of a radian -- most of a plan -- far fewer poses reach the
// narrow phase, and none that could touch is lost: the flags do not change.
//   (pocs_footprint_extent, pocs_collide.h: host + device, checked on the CPU against a dense scan)
template <int K, int TB>
__device__ __forceinline__ void gmm_cull(const pocs_gmm_launch& a, gmm_smem<K, TB>& sm, const int rb, const int lane, const double* par) {
  const pocs_footprint fp = a.fp;
  const int M = a.M;
  const double* const obs = sm.obs();
  double xlo = 1e300, xhi = -1e300, ylo = 1e300, yhi = -1e300, tlo = 1e300, thi = -1e300;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const double* p = &par[k * POCS_PARAM_STRIDE];
    const double ex = 6.67 * fabs(p[3]), ey = 6.67 * (fabs(p[4]) + fabs(p[5])), et = 6.67 * (fabs(p[6]) + fabs(p[7]) + fabs(p[8]));
    xlo = fmin(xlo, p[0] - ex); xhi = fmax(xhi, p[0] + ex);
    ylo = fmin(ylo, p[1] - ey); yhi = fmax(yhi, p[1] + ey);
    tlo = fmin(tlo, p[2] - et); thi = fmax(thi, p[2] + et);
  }
  const double pad = sqrt(fp.dx * fp.dx + fp.dy * fp.dy) + 1e-6;   // footprint centre vs base
  xlo -= pad; xhi += pad; ylo -= pad; yhi += pad;
  tlo -= 1e-9 * (1.0 + fabs(tlo)); thi += 1e-9 * (1.0 + fabs(thi));
  const double HALF_PI = 1.57079632679489661923;
  // pocs_footprint_extent_pre for world x (the range as it is) and world y (shifted by a quarter turn), with the four end
  // values -- a general sine and cosine each, ~70 dependent operations -- evaluated side by side in lanes 0 .. 3 instead of
  // one after the other in every lane: the same functions of the same arguments, a quarter of the wave's time
  const double end_t = ((lane & 1) ? thi : tlo) - ((lane & 2) ? HALF_PI : 0.0);      // tlo, thi, tlo - pi/2, thi - pi/2
  const double end_f = pocs_footprint_extent_end(fp.hx, fp.hy, end_t);
  const double ext_x = pocs_footprint_extent_is_radius(a.fp_phi, tlo, thi) ? a.fp_rr
                     : pocs_footprint_extent_of_ends(a.fp_rr, lane_value(end_f, 0), lane_value(end_f, 1));
  const double ext_y = pocs_footprint_extent_is_radius(a.fp_phi, tlo - HALF_PI, thi - HALF_PI) ? a.fp_rr
                     : pocs_footprint_extent_of_ends(a.fp_rr, lane_value(end_f, 2), lane_value(end_f, 3));
  bool keep = false;
  double bx = 0.0, by = 0.0;
  if (lane < M) {
    const double* o = &obs[lane * POCS_OBS_STRIDE];
    // the obstacle's own world box (as pocs_prepare_obstacle) + the footprint's extents for this run
    bx = fmin(o[6], fma(o[4], fabs(o[2]), o[5] * fabs(o[3])) * (1.0 + 1e-12) + ext_x);
    by = fmin(o[7], fma(o[4], fabs(o[3]), o[5] * fabs(o[2])) * (1.0 + 1e-12) + ext_y);
    keep = !(o[0] - bx > xhi || o[0] + bx < xlo || o[1] - by > yhi || o[1] + by < ylo);
  }
  const unsigned long long mask = __ballot(keep);
  if (keep) {
    const int pos = __popcll(mask & ((1ull << lane) - 1ull));
#pragma unroll
    for (int j = 0; j < 6; ++j) sm.keep[rb][pos * POCS_OBS_STRIDE + j] = obs[lane * POCS_OBS_STRIDE + j];
    sm.keep[rb][pos * POCS_OBS_STRIDE + 6] = bx;
    sm.keep[rb][pos * POCS_OBS_STRIDE + 7] = by;
  }
  if (lane == 0) sm.nkeep[rb] = __popcll(mask);
}

// ---------------------------------------------------------------------------------------------
// THE BODY: units [ta, tb) of the launch (unit t = virtual slice t mod VS of run t / VS; at most
// POCS_GMM_SUB of them, of at most two runs r0 and r0 + 1 whose parameters and culled tables are staged
// in buffers 0 and 1), as every thread of the block runs them: GM_Model::sampleNPoints
// (GM_Model.h:83-116) + checkMatrixCollisions (MCSimulator.h:241-253) + the moment sums (:592-611),
// fused, one PAIR of samples per thread and iteration.  The waves of the block do not meet in here: each
// works through the units at its own pace and leaves its wave sums in LDS (flush_unit).  What a lane adds
// up, and in which order, depends on (run, virtual slice, wave, lane) only.
// ---------------------------------------------------------------------------------------------
//   zpre / npre (the lone form, LONE_PRE): the normals of the unit's first `npre` iterations, drawn in the block's head by
//   the waves that waited there ([iteration][sample of the pair x 3][thread]); the same function of the same arguments,
//   the same bits -- a call of one run then spends its sampling phase on what depends on the mixture only
template <int K, bool STORE, int TB, bool LONE_PRE = false>
__device__ __forceinline__ void gmm_units(const pocs_gmm_launch& a, gmm_smem<K, TB>& sm, const int w, const int r0,
                                          const int ta, const int tb, const double* zpre = nullptr, const int npre = 0) {
  const pocs_tables* const s_tab = &sm.tab;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const pocs_footprint fp = a.fp;
  const int vs_mask = (1 << a.vs_shift) - 1;
  //   acc[0..8] = sums of x, y, t, xx, xy, xt, yy, yt, tt over the survivors of the component being
  //   accumulated (a wave works through the component blocks in order), nfree = their number (the wave's)
  double acc[9];
  int nfree = 0;
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = 0.0;
  // Positions inside the shard are 32-bit (the host refuses shards of 2^31 samples and more): LOCAL sample
  // i is global sample first + i, local pair lp holds local samples 2 lp, 2 lp + 1 (a.first is even,
  // checked by the host), and everything the unit loop decides -- chunk ranges, the end of a component
  // block, whole or general iteration -- is scalar integer arithmetic.
  const int count = (int)a.count;
  const int npairs = (count + 1) >> 1;
  const uint64_t pair0 = (uint64_t)(a.first >> 1);
  const double first_d = (double)a.first;
  const int wave_first = 128 * wave;                // the wave's first sample within a chunk
  // The (up to) four waves of a SIMD -- two of this block, two of the co-resident one -- are arbitrated
  // by priority, then AGE: left alone, the oldest wave of a SIMD runs ~1.7 x faster than the youngest for
  // the whole launch.  Rotating the priority with the iteration gives every wave the same share.
  // slot = which of the block's waves on this SIMD: wave v runs on SIMD v mod 4, so waves v and v + 4 share
  // one.  The second block of a CU is (observed, speed only) the one dispatched 256 blocks later.
  const int prio_slot = (TB >= 512 ? (wave >> 2) : 0) + (TB >= 512 ? 2 : 1) * (int)((blockIdx.x >> 8) & 3u);
  int prio_it = prio_slot;
  // per run (wave-uniform; reloaded when the block's range crosses into its second run)
  int rb = -1, nkeep = 0, kcur = 0, kw = 0;
  const double* s_par = nullptr;
  const double* s_keep = nullptr;
  uint64_t seed = 0;
  double cumn[K > 1 ? K - 1 : 1];                   // cumulative component counts
  double *xr = nullptr, *yr = nullptr, *tr = nullptr;
  int16_t* fr = nullptr;
  int seg_end = 0;                                  // local sample index up to which (exclusive) the samples belong to component kw and exist
  POCS_VCONST(vc_);                                 // polynomial constants held in vector registers (pocs_math.h)
  const pocs_vconst* const vc = &vc_;

  // ONE iteration = 2 * TB samples, one pair per thread.  WHOLE (compile time): the wave's 128 samples lie
  // inside component block kw and inside the shard -- every lane live, both samples of its pair exist,
  // the component is the scalar kw == kcur.  Otherwise: the general case (a block boundary inside the
  // wave, the shard's last chunk), every decision per lane.  Same arithmetic per sample either way.
  int it_unit = 0;                                  // (lone form) the iteration's number within the unit
  auto iteration = [&](auto whole_tag, const int base, const int tl, bool& first) __attribute__((always_inline)) {
    constexpr bool WHOLE = decltype(whole_tag)::value;
    switch (prio_it++ & 3) {                       // s_setprio takes an immediate
      case 0: __builtin_amdgcn_s_setprio(0); break;
      case 1: __builtin_amdgcn_s_setprio(1); break;
      case 2: __builtin_amdgcn_s_setprio(2); break;
      default: __builtin_amdgcn_s_setprio(3); break;
    }
    const int lp = base + tid;
    const bool live = WHOLE || lp < npairs;        // a lane past the end computes, masked
    double zz[2][3];
    uint32_t spare[2];
    // The seed is made opaque once per iteration: otherwise the compiler hoists all 20 Philox round
    // keys (seed + r * Weyl constants) out of the loop and pins 20 SGPRs of a register file that is
    // already spilling; recomputing them costs 2 scalar adds per round.
    uint64_t seed_it = seed;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(seed_it));
#endif
    bool drawn = false;
    if constexpr (LONE_PRE) {
      if (it_unit < npre) {                          // (scalar)
        const double* z = zpre + (size_t)it_unit * 6 * TB + tid;
#pragma unroll
        for (int q = 0; q < 3; ++q) { zz[0][q] = z[q * TB]; zz[1][q] = z[(3 + q) * TB]; }
        drawn = true;
        ++it_unit;
      }
    }
    if (!drawn) {
      POCS_TUNE_NORMALS(pocs_normal3_pair(seed_it, pair0 + (uint64_t)(unsigned)lp, (uint32_t)w, POCS_STREAM_GMM, s_tab, zz[0], zz[1], &spare[0], &spare[1], vc));
    }
    const int i0 = 2 * lp;
    const bool two = WHOLE || (live && (i0 + 1) < count);  // false only for the last sample of an odd shard
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
    }
    POCS_TUNE_COLLIDE_STATS();
    POCS_TUNE_COLLIDE(pocs_pair_collides<LONE_PRE>(xs, ys, ts, &fp, s_keep, nkeep, s_tab, vc, hits));
    if constexpr (POCS_TUNE_SKIP_MOMENTS) { POCS_TUNE_MOMENTS_ALT(); } else {
    // T1 sums over the collision-free samples of the component being accumulated:
    //   (x, y, t, x x, x y, x t, y y, y t, t t) with the products inside the fma; survivors by population count.
    if (WHOLE) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        nfree += __popcll(__ballot(!hits[h]));
        if (!hits[h]) {                             // the few lanes that collided sit this out
          const double x = xs[h], y = ys[h], t = ts[h];
          acc[0] += x; acc[1] += y; acc[2] += t;
          acc[3] = fma(x, x, acc[3]); acc[4] = fma(x, y, acc[4]); acc[5] = fma(x, t, acc[5]);
          acc[6] = fma(y, y, acc[6]); acc[7] = fma(y, t, acc[7]); acc[8] = fma(t, t, acc[8]);
        }
      }
    } else {
      // The components present in the wave are visited in increasing order (scalar loop), the previous
      // component's chains being flushed first; with ind = 1.0 for a surviving sample of the component and
      // 0.0 otherwise, (xm, ym, tm) = ind * (x, y, t) enter the sums -- a sample that does not count adds +-0
      // to every one of them, which is why the WHOLE form above gives the same bits.
      // Sample indices grow with the lane: the wave's first LIVE lane holds its first component, lane 63 its last.
      const unsigned long long live_mask = __ballot(live);
      const int klo = live_mask ? __builtin_amdgcn_readlane(ks[0], (int)__builtin_ctzll(live_mask)) : K;
      const int khi = __ballot(two) == ~0ull ? __builtin_amdgcn_readlane(ks[1], 63) : K - 1;
#pragma unroll
      for (int kk = 0; kk < K; ++kk) {
        if (kk < klo || kk > khi) continue;                                 // scalar compares
        if (kk != kcur) { flush_unit(sm, wave, lane, rb, tl, kcur, first, acc, nfree); kcur = kk; }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bool sel = (h == 0 ? live : two) && ks[h] == kk;
          const bool cnt = sel && !hits[h];
          nfree += __popcll(__ballot(cnt));
          const double ind = cnt ? 1.0 : 0.0;
          const double xm = ind * xs[h], ym = ind * ys[h], tm = ind * ts[h];
          acc[0] += xm; acc[1] += ym; acc[2] += tm;
          acc[3] = fma(xm, xs[h], acc[3]); acc[4] = fma(xm, ys[h], acc[4]); acc[5] = fma(xm, ts[h], acc[5]);
          acc[6] = fma(ym, ys[h], acc[6]); acc[7] = fma(ym, ts[h], acc[7]); acc[8] = fma(tm, ts[h], acc[8]);
        }
      }
    }
    }
    if (STORE && live) {
      // Both poses of the pair leave together.  For the last sample of an odd shard the second slot is
      // the pair's unused twin: it lands in the padding element of the run's slice (sample_stride >=
      // count + 1 then) and is never read back.  Written once, never re-read by the kernels: non-temporal,
      // past the caches.  The loop counter is wave-uniform (SGPRs) and the lane adds its tid: the store
      // addresses are a scalar base per iteration plus a constant 16*tid, no per-lane 64-bit arithmetic.
      const size_t ub = 2 * (size_t)(unsigned)base;
      const int fl = (hits[0] ? 1 : 0) | ((two && hits[1]) ? 0x10000 : 0);
      store16_nt(xr + ub, 16u * (unsigned)tid, (v2d){xs[0], xs[1]});
      store16_nt(yr + ub, 16u * (unsigned)tid, (v2d){ys[0], ys[1]});
      store16_nt(tr + ub, 16u * (unsigned)tid, (v2d){ts[0], ts[1]});
      store4_nt(fr + ub, 4u * (unsigned)tid, fl);
    }
  };

  // the component block the wave's local sample l0 lies in, and where whole waves of it end
  auto lookup = [&](const int l0) __attribute__((always_inline)) {
    const double g0 = first_d + (double)l0;
    int kk = 0;
#pragma unroll
    for (int q = 0; q < K - 1; ++q) kk += (cumn[q] <= g0) ? 1 : 0;
    kw = __builtin_amdgcn_readfirstlane(kk);
    double e = (double)count;
    if (kw < K - 1) e = fmin(e, s_par[kw * POCS_PARAM_STRIDE + 9] - first_d);      // exact: integers below 2^53
    seg_end = __builtin_amdgcn_readfirstlane((int)e);
  };
  for (int t = ta; t < tb; ++t) {
    const int tl = t - ta;
    const int r = t >> a.vs_shift, j = t & vs_mask;
    if (r - r0 != rb) {                              // (scalar) the block's first unit, or its range enters run r0 + 1
      rb = r - r0;
      s_par = sm.par[rb]; s_keep = sm.keep[rb];
      nkeep = __builtin_amdgcn_readfirstlane(sm.nkeep[rb]);
      seed = (uint64_t)uniform64((long long)sm.seed[rb]);         // scalar registers, provably
#pragma unroll
      for (int q = 0; q < K - 1; ++q) cumn[q] = s_par[q * POCS_PARAM_STRIDE + 9];
      // (Keeping the wave's component parameters in scalar registers instead of reading them at one LDS
      // address spills scalar registers: measured 5 % slower at K = 3.)
      xr = a.x + (size_t)r * a.sample_stride;          // this run's slice (sample_stride is even)
      yr = a.y + (size_t)r * a.sample_stride;
      tr = a.th + (size_t)r * a.sample_stride;
      fr = a.flags + (size_t)r * a.sample_stride;
      seg_end = 0;                                     // nothing known about this run's component blocks yet
    }
    // chunks [c_begin, c_end) of virtual slice j: a fixed cut of the run's chunks into VS = 2^vs_shift ranges
    const int c_begin = (int)(((long long)j * a.chunks) >> a.vs_shift);
    const int c_end = (int)(((long long)(j + 1) * a.chunks) >> a.vs_shift);
    bool first = true;                               // (scalar) nothing of this unit has been flushed yet
    it_unit = (t == ta) ? 0 : npre;                  // (lone form) normals drawn ahead exist for the first unit held only
    // A wave's samples only move forward within a run, so the component block found for an earlier unit still
    // holds while the wave's 128 samples end before seg_end; it is looked up again (a few vector compares)
    // only when they do not: at a block boundary, at the shard's end.
    const int end = c_end * TB;
    int base = c_begin * TB;
    if (2 * base + wave_first + 128 > seg_end) lookup(2 * base + wave_first);
    kcur = kw;                                       // the component the wave's first sample of the unit belongs to
    // A wave's 128 samples of an iteration nearly always lie inside ONE component block and inside the shard:
    // those iterations run in the inner loop below, which knows nothing of the general case (no per-lane
    // index compares, no masks, no flush; the accumulators stay where they are).
    while (base < end) {
      int l0 = 2 * base + wave_first;                                        // local index of the wave's first sample
      if (l0 + 128 > seg_end) lookup(l0);
      if (l0 + 128 <= seg_end) {
        if (kw != kcur) { flush_unit(sm, wave, lane, rb, tl, kcur, first, acc, nfree); kcur = kw; }
        do {
          iteration(std::true_type{}, base, tl, first);
          base += TB;
          l0 += 2 * TB;
        } while (base < end && l0 + 128 <= seg_end);
      } else {
        iteration(std::false_type{}, base, tl, first);
        base += TB;
      }
    }
    __builtin_amdgcn_s_setprio(3);                   // flushes run at the top priority: other waves will wait for them
    flush_unit(sm, wave, lane, rb, tl, kcur, first, acc, nfree);      // the unit's last component
  }
}

// The rows of the held units [ta, tb): column c of (virtual slice, component) = the eight wave sums in wave
// order -- whichever of them the unit has: slot if the component was the wave's first there, xtra if it
// started inside the wave's share, +0 otherwise -- stored write-through to the run's partial rows.
// Column 1 (collisions) stays 0: a component's collisions are what is left of its block (gmm_close_sums).
template <int K, int TB>
__device__ __forceinline__ void gmm_emit_rows(const pocs_gmm_launch& a, gmm_smem<K, TB>& sm, const int r0, const int ta, const int tb) {
  constexpr int NC = K * POCS_NMOM, NW = TB / 64;
  for (int i = threadIdx.x; i < (tb - ta) * NC; i += TB) {
    const int tl = i / NC, c = i - tl * NC, k = c / POCS_NMOM, col = c - k * POCS_NMOM;
    const int rb = ((ta + tl) >> a.vs_shift) - r0;
    double v = 0.0;
    if (col != 1) {
      const int s = col == 0 ? 0 : col - 1;
      // the eight waves' terms are fetched side by side (both candidates of each: one LDS round trip), then added in wave order
      double ps[NW], px[NW];
      int kf[NW], xj[NW];
#pragma unroll
      for (int u = 0; u < NW; ++u) { kf[u] = sm.kf[tl][u]; xj[u] = sm.xj[rb][u][k]; ps[u] = sm.slot[tl][u][s]; px[u] = sm.xtra[rb][u][k][s]; }
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const double p = kf[u] == k ? ps[u] : (xj[u] == tl ? px[u] : 0.0);
        v = (u == 0) ? p : v + p;
      }
    }
    store_wt(&a.partial[(size_t)(ta + tl) * NC + c], v);        // [run][VS][NC] = [unit][NC]
  }
}

// The closer of (r, w) -- the block that drew the run's last ticket, behind its acquire -- adds the VS
// partial rows of the run in a FIXED order that does not depend on who adds them: sixteen interleaved
// partial sums per column, P_g = row g + row (g + 16) + row (g + 32) + ... in that order, then
// P_0 + P_1 + ... + P_15 in that order.  Every (g, column) is one work item: its loads are L1-bypassing
// and independent, so the block has the whole table in flight at once -- ONE memory round trip.
// `stage`: >= 16 * NC doubles of LDS.  The collisions of a component are what is left of its block of
// this shard's samples: nColl_k = n_k - nFree_k, with [cum_{k-1}, cum_k) the component's global sample
// range (par[k][9], cum_{K-1} = n_total).  Result: tot[c], and moments[w][r][c] in global memory (it
// leaves the launch at the kernel boundary).
// The rows of ONE batch of work items (two per thread, starting at item i0): request() issues every load, reduce() adds
// them in row order into the staging rows.  A caller with other requests to make (the lone form's heads; the closer, whose
// mixture advance wants state[w] and the chain record) issues the first batch itself, next to them.
template <int K, int NT>
struct close_rows {
  static constexpr int NC = K * POCS_NMOM, G = 16, ITEMS = G * NC, RPI = POCS_GMM_MAX_VS / G;
  double v[2][RPI];
  __device__ __forceinline__ void request(const double* src, const int S, const int i0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = i0 + t * NT, g = i / NC, c = i - g * NC;
#pragma unroll
      for (int u = 0; u < RPI; ++u) {
        const int q = g + u * G;
        v[t][u] = (i < ITEMS && q < S) ? load_wt(&src[(size_t)q * NC + c]) : 0.0;
      }
    }
  }
  // every value passes through an (empty) asm statement: the additions of reduce() cannot rise above it into the branches
  // of the loads -- where the compiler, left to itself, puts an item's first one, with a wait for that load in front of
  // all the others
  __device__ __forceinline__ void pin() {
    static_assert(RPI % 4 == 0, "four values per statement");
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int u = 0; u < RPI; u += 4) asm volatile("" : "+v"(v[t][u]), "+v"(v[t][u + 1]), "+v"(v[t][u + 2]), "+v"(v[t][u + 3]));
  }
  __device__ __forceinline__ void reduce(double* stage, const int S, const int i0) const {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = i0 + t * NT, g = i / NC;
      double sum = 0.0;
#pragma unroll
      for (int u = 0; u < RPI; ++u) if (g + u * G < S) sum += v[t][u];
      if (i < ITEMS) stage[i] = sum;
    }
  }
};
template <int K>
__device__ __forceinline__ const double* close_rows_of(const pocs_gmm_launch& a, const double* rows, const int r) {
  return rows + (size_t)r * (1 << a.vs_shift) * (K * POCS_NMOM);
}
//   REQUESTED: the caller has issued the first batch (items tid, tid + NT) into `cr` already
template <int K, int NT, bool REQUESTED = false>
__device__ __forceinline__ void gmm_close_sums(const pocs_gmm_launch& a, const int w, const int r, const double* s_par,
                                               double* stage, double* tot, const int tid,
                                               const double* rows, const bool store, close_rows<K, NT>& cr) {
  constexpr int NC = K * POCS_NMOM, G = 16, ITEMS = G * NC, nthreads = NT;
  const int S = 1 << a.vs_shift;
  const double* src = close_rows_of<K>(a, rows, r);
  // An item's (up to) sixteen rows g, g + 16, ... are all requested before the first is waited for, and so are the
  // rows of the thread's NEXT item where there are more items than threads (K = 3: 528 items on 512 threads --
  // taken one after the other, sixteen threads cost the whole block a second memory round trip): two items per
  // batch, ONE round trip per batch for the run's 256 rows.  Added in row order.  (requests_issued() between the two
  // steps: left to itself the compiler folds an item's first addition into the branch of its first load and waits
  // there -- three round trips per batch instead of one.)
  for (int i0 = tid; i0 < ITEMS; i0 += 2 * NT) {
    if (!(REQUESTED && i0 == tid)) cr.request(src, S, i0);
    requests_issued();
    cr.pin();
    cr.reduce(stage, S, i0);
  }
  __syncthreads();
  for (int c = tid; c < NC; c += nthreads) {
    double v = stage[c];
    const int ng = S < G ? S : G;
    for (int g = 1; g < ng; ++g) v += stage[g * NC + c];
    tot[c] = v;
  }
  __syncthreads();
  const double lo = (double)a.first, hi = (double)(a.first + a.count);
  for (int k = tid; k < K; k += nthreads) {
    const double c0 = (k == 0) ? 0.0 : s_par[(k - 1) * POCS_PARAM_STRIDE + 9];
    const double c1 = (k == K - 1) ? (double)a.n_total : s_par[k * POCS_PARAM_STRIDE + 9];
    const double n_k = fmax(0.0, fmin(c1, hi) - fmax(c0, lo));
    tot[k * POCS_NMOM + 1] = n_k - tot[k * POCS_NMOM];
  }
  __syncthreads();
  if (store) for (int c = tid; c < NC; c += nthreads) a.moments[((size_t)w * a.nruns + r) * NC + c] = tot[c];
}

// One waypoint of runs [run_lo, run_lo + run_cnt) of the call as ONE launch (the host issues a call's runs as
// one launch per waypoint, or as two half-batches on two streams whose launches overlap, pocs_host.hip).
// The launch's work is the flat list of UNITS t = r * VS + j (virtual slice j of run r); block b takes the
// b-th `upb` of them -- every block
// the same number, whatever the number of runs, which is what lets a launch of ANY number of runs fill
// the 512 resident blocks evenly.  A block's range may cross from one run into the next (never further:
// upb <= VS).
//   head  log / sector tables, the obstacle table, the sampler parameters of the block's (up to) two
//         runs -> LDS; exact culling of the obstacle table per run (waves 0 and 1);
//   body  gmm_units, POCS_GMM_SUB units at a time, each followed by the rows of those units -> the
//         write-through partial rows [run][VS];
//   tail  every storing wave drains -> the block meets -> one ticket per run it touched; the block that
//         draws a run's last ticket acquires, adds the run's VS rows and (one GPU) advances the mixture
//         to the next waypoint -- sharded: after exchanging the run's moments with the other ranks.
//
// LONE (one run per call, no batch, no run-ahead): nothing else is in flight to hide a closer behind, and the
// tickets' two round trips (drain the rows, draw the ticket) and the closer's acquire are pure latency.  The
// launch of waypoint w then closes waypoint w - 1 ITSELF, in the head of EVERY block: the rows of w - 1 (the
// other half of the row buffer: a fast block must not overwrite what a slow one still reads), state[w-1] and
// the chain record arrive in one round trip behind the kernel boundary; every block adds the rows in the
// fixed order, advances the mixture -- 256 times the same few microseconds of one wave, on CUs that would
// otherwise wait for one of them to do it -- and keeps param[w] in LDS; block 0 writes moments[w-1], state[w],
// param[w] out for the getters.  The tail is the rows' stores and nothing else; a one-block launch
// (k_gmm_close) adds the last waypoint's rows.  Same functions, same order of additions: the same bits.
template <int K, bool STORE, int TB, bool LONE>
__global__ __launch_bounds__(TB, (LONE ? 1 : POCS_GMM_BLOCKS_PER_CU) * TB / 256) void k_gmm_step(pocs_gmm_launch a) {   // (lone: one block per CU, its LDS)
  typedef gmm_smem<K, TB> smem_t;
  constexpr int SUB = smem_t::SUB, NW = smem_t::NW;
  __shared__ smem_t sm;
  // (lone form: one block per CU, the whole LDS is its own) the normals of the unit's first POCS_LONE_PRE iterations
  __shared__ double s_zpre[LONE ? POCS_LONE_PRE * 6 * TB : 2];
  int npre = 0;
  const int tid = threadIdx.x;
  const int w = a.waypoint;
  const int t_lo = a.run_lo << a.vs_shift, t_hi = (a.run_lo + a.run_cnt) << a.vs_shift;    // this launch's units
  const int bx = (int)blockIdx.x;
  const int t0 = t_lo + bx * a.upb;
  const int t1 = (t0 + a.upb < t_hi) ? t0 + a.upb : t_hi;
  const int r0 = t0 >> a.vs_shift, r1 = (t1 - 1) >> a.vs_shift;       // the block's first and last run (r1 <= r0 + 1)
  POCS_STAMP_BEGIN();
  // The head's inputs are all REQUESTED before the first of them is waited for: the tables (24 bytes per thread), and per
  // form what follows -- one memory round trip for the lot, then the stores to LDS.
  constexpr int PS = K * POCS_PARAM_STRIDE;
  static_assert(POCS_MAX_OBSTACLES * POCS_OBS_STRIDE <= TB, "one obstacle element per thread");
  static_assert(2 * PS <= TB, "one sampler parameter per thread (two runs)");
  static_assert(POCS_ADV_STAGE_MAX <= 4 * TB, "the advance's inputs in one batch");
  double tabv[table_regs<TB>::N];
  request_tables<TB>(a.tables, tid, tabv);
  const double obs_elem = tid < a.M * POCS_OBS_STRIDE ? a.env->obs[tid] : 0.0;
  if (LONE && w > 0) {
    // close waypoint w - 1 and advance to w, here (r0 is the call's one run)
    const bool out = blockIdx.x == 0;
    const adv_ptrs ap = advance_ptrs(a, K, w, r0, sm.adv());
    double advv[4];
    advance_request(ap, false, tid, TB, advv);                                             // state[w-1], chain record, sensor (the moments come from the rows)
    const double parv = tid < PS ? a.param[((size_t)r0 * a.W + (w - 1)) * PS + tid] : 0.0;   // (the counts of w - 1)
    const unsigned long long seedv = a.hdr[r0].seed;
    close_rows<K, TB> cr;
    cr.request(close_rows_of<K>(a, a.partial_prev, r0), 1 << a.vs_shift, tid);             // the rows of w - 1
    requests_issued();
    commit_tables<TB>(&sm.tab, tid, tabv);
    advance_commit(ap, false, tid, TB, advv);
    if (tid < PS) sm.par[1][tid] = parv;
    for (int j = tid; j < 2 * NW * K; j += TB) (&sm.xj[0][0][0])[j] = -1;
    if (tid == 0) sm.seed[0] = seedv;
    double* const l_mom = ap.l_mom;
    // (the obstacle table, one element per thread, requested with everything else: it lands in the transpose scratch as soon
    // as the row sums are done with it, under the components' serial chain instead of in a round trip of its own behind it)
    gmm_close_sums<K, TB, true>(a, w - 1, r0, sm.par[1], sm.stage(), l_mom, tid, a.partial_prev, out, cr);
    POCS_STAMP(5);
    if (tid < a.M * POCS_OBS_STRIDE) sm.obs()[tid] = obs_elem;      // (every read of the staging rows lies behind a barrier of gmm_close_sums)
    // While wave 0 walks the components' serial chain and a lane of wave 1 draws the counts, the other six waves draw
    // the normals of the block's first unit -- they do not depend on the mixture -- for all 512 threads and the unit's
    // first iterations: (seed, pair index, waypoint) -> six normals, the arguments the sampling loop would use.
    const int j0 = t0 & ((1 << a.vs_shift) - 1);
    const int cb0 = (int)(((long long)j0 * a.chunks) >> a.vs_shift), ce0 = (int)(((long long)(j0 + 1) * a.chunks) >> a.vs_shift);
    npre = (ce0 - cb0) < POCS_LONE_PRE ? (ce0 - cb0) : POCS_LONE_PRE;
    auto draw_ahead = [&]() __attribute__((always_inline)) {
      POCS_VCONST(vc_);
      const uint64_t seed = seedv;                          // (requested with the head's other inputs)
      const uint64_t pair0 = (uint64_t)(a.first >> 1);
      for (int it = 0; it < npre; ++it)
        for (int l = tid - 128; l < TB; l += TB - 128) {
          double za[3], zb[3];
          uint32_t sa, sb;
          pocs_normal3_pair(seed, pair0 + (uint64_t)(unsigned)((cb0 + it) * TB + l), (uint32_t)w, POCS_STREAM_GMM, &sm.tab, za, zb, &sa, &sb, &vc_);
          double* z = s_zpre + (size_t)it * 6 * TB + l;
#pragma unroll
          for (int q = 0; q < 3; ++q) { z[q * TB] = za[q]; z[(3 + q) * TB] = zb[q]; }
        }
    };
    // (the cull of the obstacle table against the mixture of waypoint w -- means and factors only -- by wave 1, beside wave 0's
    // normalisation and publishing instead of behind them and a barrier)
    auto cull_early = [&]() __attribute__((always_inline)) { gmm_cull(a, sm, 0, tid - 64, ap.l_par); };
    if (tid < 128) __builtin_amdgcn_s_setprio(3);          // the components' chain and the count lane go first on their SIMDs; the drawing waves fill in
    advance_block(a, K, w, r0, sm.adv(), sm.spec(), true, tid, TB, true, out, draw_ahead, cull_early);
    if (tid < 128) __builtin_amdgcn_s_setprio(0);
    __syncthreads();
    POCS_STAMP(6);
    POCS_STAMP_COUNT(14);
    const double* const l_par = advance_ptrs(a, K, w, r0, sm.adv()).l_par;
    for (int j = tid; j < PS; j += TB) sm.par[0][j] = l_par[j];
  } else {
    // param[r][w][..]: the two runs' records are a.W records apart
    const int rp = tid / PS, jp = tid - rp * PS;
    const double parv = tid < (r1 - r0 + 1) * PS ? load_wt(&a.param[((size_t)(r0 + rp) * a.W + w) * PS + jp]) : 0.0;
    const unsigned long long seedv = tid <= r1 - r0 ? a.hdr[r0 + tid].seed : 0ull;
    requests_issued();
    commit_tables<TB>(&sm.tab, tid, tabv);
    if (tid < a.M * POCS_OBS_STRIDE) sm.obs()[tid] = obs_elem;
    if (tid < (r1 - r0 + 1) * PS) sm.par[rp][jp] = parv;
    for (int j = tid; j < 2 * NW * K; j += TB) (&sm.xj[0][0][0])[j] = -1;
    if (tid <= r1 - r0) sm.seed[tid] = seedv;
  }
  __syncthreads();
  if (!(LONE && w > 0)) {                            // (scalar; the lone form's heads have culled already, above)
    if (tid < 64) gmm_cull(a, sm, 0, tid, sm.par[0]);
    else if (tid < 128 && r1 > r0) gmm_cull(a, sm, 1, tid - 64, sm.par[1]);
    __syncthreads();
  }                                                  // from here on the transpose scratch is the waves'
  POCS_STAMP(0);
  for (int ta = t0; ta < t1; ta += SUB) {
    const int tb = (ta + SUB < t1) ? ta + SUB : t1;
    gmm_units<K, STORE, TB, LONE>(a, sm, w, r0, ta, tb, s_zpre, ta == t0 ? npre : 0);
    POCS_STAMP(1);
    __syncthreads();
    POCS_STAMP(2);
    gmm_emit_rows(a, sm, r0, ta, tb);
    if (tb < t1) {                                   // more units to come (only launches of > 64 runs): the slots start over
      __syncthreads();
      for (int j = tid; j < 2 * NW * K; j += TB) (&sm.xj[0][0][0])[j] = -1;
      __syncthreads();
    }
  }
  if (LONE) return;                                  // the rows leave through the kernel boundary; the next launch's heads add them
  drain_stores();
  __syncthreads();
  POCS_STAMP(3);
  if (tid <= r1 - r0) {                              // one ticket per run touched: the blocks whose range meets [r VS, (r + 1) VS)
    const int r = r0 + tid;
    const int b_first = ((r << a.vs_shift) - t_lo) / a.upb, b_last_raw = ((((r + 1) << a.vs_shift) - 1) - t_lo) / a.upb;
    const int b_last = b_last_raw < (int)gridDim.x - 1 ? b_last_raw : (int)gridDim.x - 1;
    const unsigned t = __hip_atomic_fetch_add(&a.ticket[(size_t)r * a.W + w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sm.last[tid] = (t == (unsigned)(b_last - b_first)) ? 1 : 0;
  }
  if (tid == 0 && r1 == r0) sm.last[1] = 0;
  __syncthreads();
  POCS_STAMP(4);
  // The closer of a run, written out for the block's first run and -- when its range crosses into a second one --
  // once more, NOT as a loop over the two: the advance wants every vector register there is, and in a loop
  // whatever the compiler hoists out of the body (thread-dependent addresses) is live across it and spilled
  // (256 bytes of scratch per lane instead of 60).
  auto closer = [&](const int rb) __attribute__((always_inline)) -> bool {
    const int r = r0 + rb;
    if (tid == 0) acquire_agent();
    __syncthreads();
    // the run's rows and -- one GPU -- what the mixture advance wants besides their sums (state[w], the chain record, the
    // sensor: written by earlier launches), requested together: one round trip instead of one after the other
    const adv_ptrs ap = advance_ptrs(a, K, w + 1, r, sm.adv());
    double* const l_mom = ap.l_mom;
    double advv[4];
    advance_request<true>(ap, false, tid, TB, advv, a.advance_in_tail != 0);
    close_rows<K, TB> cr;
    cr.request(close_rows_of<K>(a, a.partial, r), 1 << a.vs_shift, tid);
    requests_issued();
    if (a.advance_in_tail) advance_commit(ap, false, tid, TB, advv);
    gmm_close_sums<K, TB, true>(a, w, r, sm.par[rb], sm.stage(), l_mom, tid, a.partial, true, cr);
    POCS_STAMP(5);
    if (a.exchange_in_tail) {
      // sharded: the run's closer is also its messenger -- this shard's moments go to every rank, the world's
      // come back summed in rank order, and the mixture advances here (no launch, no host, between waypoints)
      __syncthreads();
      if (__hip_atomic_load(&a.sync[POCS_SYNC_ABORT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      // (a whole call replayed from a graph bakes its arguments in: the call's number -- part of every row's epoch and
      // of the choice between the two slot sets -- then travels in the run's header, uploaded per call like its seed)
      unsigned long long epoch = a.xchg.epoch;
      int parity = a.xchg.parity;
      if (a.xchg_epoch_from_header) {
        const unsigned long long calls = a.hdr[r].pad;
        epoch = (calls << 20) | (unsigned long long)(w + 1);
        parity = (int)((calls * (unsigned long long)a.W + (unsigned long long)w) & 1ull);
      }
      if (!gmm_exchange_rows(a, a.xchg, epoch, parity, K, w, r, l_mom, l_mom, tid, TB, &sm.nkeep[0] /* free by now */)) return false;
      __syncthreads();                                 // the world's sums are in l_mom for the advance (which no longer starts with a staging barrier)
    }
    if (a.advance_in_tail) advance_block(a, K, w + 1, r, sm.adv(), sm.spec(), true, tid, TB, true);     // (staged above; gmm_close_sums' barriers lie between)
    __syncthreads();
    POCS_STAMP(6);
    POCS_STAMP_COUNT(14);
    return true;
  };
  const int last0 = __builtin_amdgcn_readfirstlane(sm.last[0]), last1 = __builtin_amdgcn_readfirstlane(sm.last[1]);
  if (last0 && !closer(0)) return;
  if (last1) (void)closer(1);
}

// Lone call, behind the last waypoint's launch: its rows -> moments[W-1] (one block).
template <int K>
__global__ __launch_bounds__(256) void k_gmm_close(pocs_gmm_launch a) {
  constexpr int NC = K * POCS_NMOM, PS = K * POCS_PARAM_STRIDE;
  __shared__ double s_par[PS];
  __shared__ double s_stage[16 * NC];
  __shared__ double s_tot[NC];
  const int w = a.waypoint, r = a.run_lo;
  for (int j = threadIdx.x; j < PS; j += 256) s_par[j] = a.param[((size_t)r * a.W + w) * PS + j];
  close_rows<K, 256> cr;
  gmm_close_sums<K, 256>(a, w, r, s_par, s_stage, s_tot, threadIdx.x, a.partial, true, cr);    // (first barrier: s_par is in)
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
  stage_mc_head(a.env, a.tables, s_obs, &s_fp, &s_M, &s_tab);
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
  stage_mc_head(a.env, a.tables, s_obs, &s_fp, &s_M, &s_tab);
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
  stage_mc_head(a.env, a.tables, s_obs, &s_fp, &s_M, &s_tab);
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
  if (a.lone) {
    if (a.store) hipLaunchKernelGGL((k_gmm_step<K, true, TB, true>), dim3(a.blocks), dim3(TB), 0, s, a);
    else         hipLaunchKernelGGL((k_gmm_step<K, false, TB, true>), dim3(a.blocks), dim3(TB), 0, s, a);
  } else {
    if (a.store) hipLaunchKernelGGL((k_gmm_step<K, true, TB, false>), dim3(a.blocks), dim3(TB), 0, s, a);
    else         hipLaunchKernelGGL((k_gmm_step<K, false, TB, false>), dim3(a.blocks), dim3(TB), 0, s, a);
  }
  return hipGetLastError();
}
template <int K>
hipError_t launch_gmm_close_k(const pocs_gmm_launch& a, hipStream_t s) {
  hipLaunchKernelGGL((k_gmm_close<K>), dim3(1), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace

#ifdef POCS_TUNING
#define POCS_TUNING_REPORT
#include "pocs_tuning.h"
#endif

hipError_t pocs_launch_gmm_close(int K, const pocs_gmm_launch& a, hipStream_t s) {
  switch (K) {
    case 1: return launch_gmm_close_k<1>(a, s);
    case 2: return launch_gmm_close_k<2>(a, s);
    case 3: return launch_gmm_close_k<3>(a, s);
    case 4: return launch_gmm_close_k<4>(a, s);
    case 5: return launch_gmm_close_k<5>(a, s);
    case 6: return launch_gmm_close_k<6>(a, s);
    case 7: return launch_gmm_close_k<7>(a, s);
    case 8: return launch_gmm_close_k<8>(a, s);
    default: return hipErrorInvalidValue;
  }
}

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
// The hot path's table-driven sampler functions on words / angles the caller picks (pocs_probe_device_math: a test
// hook -- the words a free-running launch meets once in 2^32 draws, the cells' edges, a heading on a sector's tie):
// the same inline functions, the tables staged in LDS as k_gmm_step stages them, the constants pinned as there.
__global__ __launch_bounds__(POCS_BLOCK) void k_probe_math(const pocs_tables* __restrict__ tables, int n, const uint32_t* __restrict__ wr,
                                                          const uint32_t* __restrict__ wa, const double* __restrict__ x,
                                                          double* __restrict__ out) {
  __shared__ pocs_tables s_tab;
  stage_tables(tables, &s_tab);
  __syncthreads();
  POCS_VCONST(vc_);
  for (int i = threadIdx.x; i < n; i += POCS_BLOCK) {
    double z0, z1, sn, cs;
    pocs_normal_pair_w2(wr[i], wa[i], &s_tab, &z0, &z1, &vc_);
    pocs_sincos_tab(x[i], &s_tab, &sn, &cs, &vc_);
    out[i] = z0; out[n + i] = z1; out[2 * n + i] = sn; out[3 * n + i] = cs;
    out[4 * n + i] = pocs_radius2_unit32(wr[i], &s_tab);
  }
}
hipError_t pocs_launch_probe_math(const pocs_tables* tables, int n, const uint32_t* wr, const uint32_t* wa, const double* x, double* out, hipStream_t s) {
  hipLaunchKernelGGL(k_probe_math, dim3(1), dim3(POCS_BLOCK), 0, s, tables, n, wr, wa, x, out);
  return hipGetLastError();
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
