// pocs_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the hot path.
//
//   k_gmm_step      S1+C1+T1  one waypoint of truncateGMM (MCSimulator.h:570-642) in ONE launch, for
//                             every run of a batch of independent estimations (blockIdx.y = run):
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
#if defined(POCS_TRACE_PHASES)
#include <stdio.h>
#endif

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

// Relaxed agent-scope accesses = write-through / L1-bypassing (sc1) on gfx950: what the
// last-arriver hand-off of the block partials uses instead of a release/acquire fence pair.
__device__ __forceinline__ void store_wt(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_wt(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(
      reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Mixture bookkeeping of waypoint `w` (see pocs_gmm_advance_component), run by ONE wave: all 64
// lanes first pull every input (state[w-1], moments[w-1], the chain record of step w-1, the
// sensor) into LDS in one round trip, lanes < K then take one component each, lane 0 normalises,
// and the wave writes state[w] / param[w] back together -- two global round trips instead of one
// per dependent access.  Called by the last block of k_gmm_step (single GPU) or by k_gmm_advance
// (waypoint 0, and after the all-reduce when sharded).
#define POCS_ADV_SCRATCH(K) ((K) * (2 * POCS_STATE_STRIDE + POCS_NMOM + POCS_PARAM_STRIDE) + POCS_CHAIN_STRIDE + \
                             (int)(sizeof(pocs_sensor) / sizeof(double)))
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

// first half: inputs -> LDS, one component per lane through truncation, EKF and Cholesky
__device__ __forceinline__ void advance_components(const pocs_gmm_launch& a, int K, int w, int r, int lane, double* scratch) {
  const adv_ptrs p = advance_ptrs(a, K, w, r, scratch);
  constexpr int SEN = (int)(sizeof(pocs_sensor) / sizeof(double));
  for (int j = lane; j < p.ss; j += 64) p.l_prev[j] = p.g_prev[j];
  if (w > 0) for (int j = lane; j < p.NC; j += 64) p.l_mom[j] = p.g_mom[j];
  for (int j = lane; j < POCS_CHAIN_STRIDE; j += 64) p.l_ch[j] = p.g_ch[j];
  for (int j = lane; j < SEN; j += 64) p.l_sen[j] = p.g_sen[j];
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
  if (lane < K)
    pocs_gmm_advance_component(lane, p.l_prev, (w == 0) ? nullptr : p.l_mom, p.l_ch, p.l_ch + 3, p.l_ch + POCS_CHAIN_Z,
                               reinterpret_cast<const pocs_sensor*>(p.l_sen), p.l_next, p.l_par);
  __threadfence_block();
  __builtin_amdgcn_wave_barrier();
}

// The component counts of waypoint w as they will come out unless a Cholesky factorisation fails
// in advance_components (which nobody can know before it has run): one lane of ANOTHER wave draws
// them while the EKF lanes work.  spec = K cumulative counts, then K alive flags assumed.
#define POCS_SPEC_SCRATCH(K) ((K) * (POCS_STATE_STRIDE + 2))
__device__ __forceinline__ void speculate_counts(const pocs_gmm_launch& a, int K, int w, int r, double* spec) {
  const adv_ptrs p = advance_ptrs(a, K, w, r, nullptr);
  double* st = spec + 2 * K;                                   // a K x STATE_STRIDE image: only [12], [13] matter
  for (int k = 0; k < K; ++k) {
    const double alive_prev = p.g_prev[k * POCS_STATE_STRIDE + 13];
    const double n = p.g_mom[k * POCS_NMOM];
    const bool alive = alive_prev != 0.0 && n >= 2.0;          // pocs_gmm_advance_component / pocs_truncated_moments
    st[k * POCS_STATE_STRIDE + 12] = alive ? n : 0.0;
    st[k * POCS_STATE_STRIDE + 13] = alive ? alive_prev : 0.0;
    spec[K + k] = st[k * POCS_STATE_STRIDE + 13];
  }
  const int last_alive = pocs_normalise_weights(K, 1, st);
  pocs_component_counts(K, st, last_alive, a.hdr[r].seed, (uint32_t)w, (double)a.n_total, spec, 1);
}

// second half: weights, component counts (the speculated ones if their premise held), write back
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
  for (int j = lane; j < p.ss; j += 64) p.g_state[(size_t)w * p.ss + j] = p.l_next[j];
  for (int j = lane; j < p.ps; j += 64) p.g_param[(size_t)w * p.ps + j] = p.l_par[j];
}

__global__ __launch_bounds__(128) void k_gmm_advance(pocs_gmm_launch a, int K) {
  __shared__ double s_adv[POCS_ADV_SCRATCH(POCS_MAX_GAUSSIANS)];
  __shared__ double s_spec[POCS_SPEC_SCRATCH(POCS_MAX_GAUSSIANS)];
  const int tid = threadIdx.x, w = a.waypoint, r = blockIdx.x;      // one block per run
  if (tid < 64) advance_components(a, K, w, r, tid, s_adv);
  else if (tid == 64 && w > 0) speculate_counts(a, K, w, r, s_spec);  // meanwhile, on the second wave
  __syncthreads();
  if (tid < 64) advance_finish(a, K, w, r, tid, s_adv, w > 0 ? s_spec : nullptr);
}

#if defined(POCS_TRACE_PHASES)     // timing-only build (tools/fixed_cost.py): 100 MHz timestamps per phase
__device__ unsigned long long g_phase[64];
#define POCS_PHASE(i) do { if (threadIdx.x == 0 && blockIdx.y == 0) ph[i] = wall_clock64(); } while (0)
#else
#define POCS_PHASE(i) do { } while (0)
#endif

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

template <int K, bool STORE, int TB>
__global__ __launch_bounds__(TB) void k_gmm_step(pocs_gmm_launch a) {
#if defined(POCS_TRACE_PHASES)
  unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  POCS_PHASE(0);
  constexpr int NC = K * POCS_NMOM;
  __shared__ double s_obs[POCS_MAX_OBSTACLES * POCS_OBS_STRIDE];
  __shared__ double s_par[K * POCS_PARAM_STRIDE];
  __shared__ double s_red[TB / 16][NC];     // one row of sums per 16-lane DPP row
  __shared__ double s_part[TB];
  __shared__ double s_adv[POCS_ADV_SCRATCH(K)];
  __shared__ double s_spec[POCS_SPEC_SCRATCH(K)];
  __shared__ double s_keep[POCS_MAX_OBSTACLES * POCS_OBS_STRIDE];
  __shared__ pocs_tables s_tab;
  __shared__ int s_nkeep;
  __shared__ int s_last;

  const int tid = threadIdx.x;
  const int w = a.waypoint;
  const int r = blockIdx.y;                 // run of the batch (independent estimations in lockstep)
  const int M = a.M;
  const pocs_footprint fp = a.fp;

  // ---- head: stage the tables, the obstacle table and this waypoint's sampler parameters in LDS
  stage_tables(a.tables, &s_tab);
  for (int j = tid; j < M * POCS_OBS_STRIDE; j += TB) s_obs[j] = a.env->obs[j];
  for (int j = tid; j < K * POCS_PARAM_STRIDE; j += TB)
    s_par[j] = a.param[((size_t)r * a.W + w) * (K * POCS_PARAM_STRIDE) + j];
  const uint64_t seed = a.hdr[r].seed;

  // Samples come in component blocks and a thread's sample indices only grow, so a wave works
  // through the components in order: ONE set of sums per thread (the wave's current component),
  // folded into the block's LDS rows when the wave moves on to the next component.
  double acc[9];
  unsigned nfree = 0u, ncoll = 0u;
  int kcur = 0;                                      // wave-uniform
#pragma unroll
  for (int j = 0; j < 9; ++j) acc[j] = 0.0;
  for (int j = tid; j < (TB / 16) * NC; j += TB) (&s_red[0][0])[j] = 0.0;
  __syncthreads();
  POCS_PHASE(1);

  // ---- head, part 2: cull the obstacle table against the mixture's bounding box.  A Box-Muller
  // normal is bounded: u >= 2^-32 gives |z| <= sqrt(64 ln 2) < 6.661 (pocs_normal_pair_w2; 6.67 leaves
  // 0.1 % for the rounding of radius * cos), so every pose this launch can draw lies within
  // mean_k +- 6.67 (|L00|, |L10|+|L11|) of some component; an obstacle whose
  // inflated box (the broad phase of pocs_box_hit) misses that region is rejected by the broad
  // phase for every sample, so dropping it here changes no flag.
  if (tid < 64) {
    double xlo = 1e300, xhi = -1e300, ylo = 1e300, yhi = -1e300;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const double* p = &s_par[k * POCS_PARAM_STRIDE];
      const double ex = 6.67 * fabs(p[3]), ey = 6.67 * (fabs(p[4]) + fabs(p[5]));
      xlo = fmin(xlo, p[0] - ex); xhi = fmax(xhi, p[0] + ex);
      ylo = fmin(ylo, p[1] - ey); yhi = fmax(yhi, p[1] + ey);
    }
    const double pad = sqrt(fp.dx * fp.dx + fp.dy * fp.dy) + 1e-6;   // footprint centre vs base
    xlo -= pad; xhi += pad; ylo -= pad; yhi += pad;
    bool keep = false;
    if (tid < M) {
      const double* o = &s_obs[tid * POCS_OBS_STRIDE];
      keep = !(o[0] - o[6] > xhi || o[0] + o[6] < xlo || o[1] - o[7] > yhi || o[1] + o[7] < ylo);
    }
    const unsigned long long mask = __ballot(keep);
    if (keep) {
      const int pos = __popcll(mask & ((1ull << tid) - 1ull));
#pragma unroll
      for (int j = 0; j < POCS_OBS_STRIDE; ++j) s_keep[pos * POCS_OBS_STRIDE + j] = s_obs[tid * POCS_OBS_STRIDE + j];
    }
    if (tid == 0) s_nkeep = __popcll(mask);
  }
  __syncthreads();
  const int nkeep = s_nkeep;
  POCS_PHASE(2);
#if defined(POCS_TRACE_PHASES)
  if (threadIdx.x == 0 && blockIdx.y == 0) ph[10] = __builtin_readcyclecounter();
#endif

  // ---- body: one PAIR of samples (2j, 2j+1) per thread and iteration -- the pair shares two
  // Philox draws = three Box-Muller pairs (pocs_normal3_pair) and its poses leave as 16-byte stores.
  // a.first is even (checked by the host), so local sample 2*lp is global sample first + 2*lp.
  const long long stride = (long long)gridDim.x * TB;
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
  for (long long base = (long long)blockIdx.x * TB; base < npairs; base += stride) {
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
    pocs_normal3_pair(seed_it, pair0 + (uint64_t)lp, (uint32_t)w, POCS_STREAM_GMM, &s_tab, zz[0], zz[1], &spare[0], &spare[1]);
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
      const bool hit = pocs_pose_collides(x, y, t, &fp, s_keep, nkeep, &s_tab);
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
          flush_component<TB, NC>(s_red, kcur, acc, nfree, ncoll, tid);
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
      typedef double v2d __attribute__((ext_vector_type(2)));
      const size_t ub = 2 * (size_t)base;
      __builtin_nontemporal_store((v2d){xs[0], xs[1]}, reinterpret_cast<v2d*>(xr + ub) + tid);
      __builtin_nontemporal_store((v2d){ys[0], ys[1]}, reinterpret_cast<v2d*>(yr + ub) + tid);
      __builtin_nontemporal_store((v2d){ts[0], ts[1]}, reinterpret_cast<v2d*>(tr + ub) + tid);
      __builtin_nontemporal_store((hits[0] ? 1 : 0) | ((two && hits[1]) ? 0x10000 : 0), reinterpret_cast<int*>(fr + ub) + tid);
    }
  }

  POCS_PHASE(3);
#if defined(POCS_TRACE_PHASES)
  if (threadIdx.x == 0 && blockIdx.y == 0) ph[11] = __builtin_readcyclecounter() - ph[10];
#endif
  // ---- tail: the last component's sums -> LDS rows; then a fixed-order sum over the TB/16 rows
  flush_component<TB, NC>(s_red, kcur, acc, nfree, ncoll, tid);
  __syncthreads();
  POCS_PHASE(4);
  if (tid < NC) {
    double v = s_red[0][tid];
#pragma unroll 8
    for (int q = 1; q < TB / 16; ++q) v += s_red[q][tid];
    store_wt(&a.partial[((size_t)r * gridDim.x + blockIdx.x) * NC + tid], v);
  }
  // hand-off: every storing wave drains its stores, the block meets, ONE lane takes a ticket;
  // the block that draws the last ticket reads every row back (L1-bypassing loads) and adds them
  // in a fixed order: slice q of column c sums rows q, q+S, q+2S, ...; then slices in order.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  POCS_PHASE(5);
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(&a.ticket[(size_t)r * a.W + w], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = (t == gridDim.x - 1u) ? 1 : 0;
  }
  __syncthreads();
  POCS_PHASE(6);
#if defined(POCS_TRACE_PHASES)
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x == 0 && a.waypoint == 5) for (int i = 0; i < 12; ++i) g_phase[32 + i] = ph[i];
#endif
  if (s_last) {
    constexpr int S = TB / NC;
    const int q = tid / NC, c = tid - q * NC;
    double v = 0.0;
    if (q < S) {
      const int nb = (int)gridDim.x;
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
    s_part[tid] = v;
    __syncthreads();
    POCS_PHASE(7);
    if (tid < NC) {
      double tot = s_part[tid];
      for (int sl = 1; sl < S; ++sl) tot += s_part[sl * NC + tid];
      a.moments[((size_t)w * a.nruns + r) * NC + tid] = tot;
    }
    // single GPU: these ARE the global moments, so carry the mixture to the next waypoint right
    // here (one wave; the other 255 CUs are already idle) instead of paying another launch
    if (a.advance_in_tail) {
      __syncthreads();
      POCS_PHASE(8);
      if (tid < 64) advance_components(a, K, w + 1, r, tid, s_adv);
      else if (tid == 64) speculate_counts(a, K, w + 1, r, s_spec);       // meanwhile, on another wave
      __syncthreads();
      if (tid < 64) advance_finish(a, K, w + 1, r, tid, s_adv, s_spec);
    }
    POCS_PHASE(9);
#if defined(POCS_TRACE_PHASES)
    if (tid == 0 && blockIdx.y == 0 && a.waypoint == 5) { for (int i = 0; i < 12; ++i) g_phase[i] = ph[i]; g_phase[12] = blockIdx.x; }
#endif
  }
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
hipError_t launch_gmm_k(int nblk, const pocs_gmm_launch& a, hipStream_t s) {
  constexpr int TB = POCS_GMM_BLOCK_OF(K);
  if (a.store) hipLaunchKernelGGL((k_gmm_step<K, true, TB>), dim3(nblk, a.nruns), dim3(TB), 0, s, a);
  else         hipLaunchKernelGGL((k_gmm_step<K, false, TB>), dim3(nblk, a.nruns), dim3(TB), 0, s, a);
  return hipGetLastError();
}

}  // namespace

hipError_t pocs_launch_gmm_step(int K, int nblk, const pocs_gmm_launch& a, hipStream_t s) {
#if defined(POCS_TRACE_PHASES)
  if (a.waypoint == 6) {         // waypoint 5 has run: print its phase stamps (10 ns ticks), once per call
    unsigned long long h[64];
    hipStreamSynchronize(s);
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)) == hipSuccess) {
      fprintf(stderr, "phases last block (bx=%llu) us:", h[12]);
      for (int i = 1; i < 10; ++i) fprintf(stderr, " %d:%.2f", i, 0.01 * (double)(long long)(h[i] - h[0]));
      fprintf(stderr, "\nphases block 0 us:");
      for (int i = 1; i < 7; ++i) fprintf(stderr, " %d:%.2f", i, 0.01 * (double)(long long)(h[32 + i] - h[32]));
      fprintf(stderr, "  (block0 start - last start %.2f)  body: %llu shader cycles in %.2f us = %.0f MHz\n",
              0.01 * (double)(long long)(h[32] - h[0]), h[11], 0.01 * (double)(long long)(h[3] - h[2]),
              (double)h[11] / (0.01 * (double)(long long)(h[3] - h[2])));
    }
  }
#endif
  switch (K) {
    case 1: return launch_gmm_k<1>(nblk, a, s);
    case 2: return launch_gmm_k<2>(nblk, a, s);
    case 3: return launch_gmm_k<3>(nblk, a, s);
    case 4: return launch_gmm_k<4>(nblk, a, s);
    case 5: return launch_gmm_k<5>(nblk, a, s);
    case 6: return launch_gmm_k<6>(nblk, a, s);
    case 7: return launch_gmm_k<7>(nblk, a, s);
    case 8: return launch_gmm_k<8>(nblk, a, s);
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
