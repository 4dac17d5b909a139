// pocs_kernels.h -- launch interface between the host runtime (pocs_host.hip) and the gfx950
// kernels (pocs_kernels.hip).  Internal; the public boundary is include/pocs.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pocs_collide.h"
#include "pocs_model.h"

#define POCS_BLOCK 256        // MC kernels
// GMM kernels: TWO blocks of 512 threads per CU = four waves per SIMD at <= 128 VGPRs.  The sampling body
// keeps one component's sums per thread whatever K is, so one shape serves every K (measured at K = 8,
// 10^7 samples, 16 runs: 2 x 512 0.39 of the HBM peak, 1 x 768 and 3 x 256 0.40, 2 x 384 0.31).  Two
// co-resident blocks instead of one fat one: while one block is in its head or tail (parameter
// staging, block reduction, the drain of its stores, ticket, mixture advance) the other block's waves
// have the CU's issue slots.
#define POCS_GMM_BLOCK_OF(K) 512
#define POCS_GMM_BLOCKS_PER_CU 2
#define POCS_NUM_CUS 256
#define POCS_MAX_BLOCKS 2048
// Task geometry of k_gmm_step.  A chunk = one iteration of a block = POCS_GMM_BLOCK_OF(K) pairs of samples.
// A run's chunks are cut into VS = 2^vs_shift VIRTUAL SLICES, slice j = chunks [j * chunks / VS,
// (j + 1) * chunks / VS): VS depends on the shard's sample count only (the largest power of two <= min(256,
// chunks)), never on how many runs share the launch -- the moment sums are defined on the virtual slices
// (pocs_kernels.hip, "summation tree"), which is what makes a run's result independent of the batch.
// The launch's work = the flat list of units t = r * VS + j; block b takes units [b * upb, (b + 1) * upb).
#define POCS_GMM_MAX_VS 256
// Block size and slice count are part of the NUMERICS (the summation tree is defined on chunks of 512 pairs, 64-lane
// waves and at most 256 virtual slices; oracle/pocs_oracle.c::tree_moments restates exactly these): a build with
// other values would produce other last bits under the same version string.
static_assert(POCS_GMM_BLOCK_OF(3) == 512 && POCS_GMM_MAX_VS == 256, "summation tree (numerics v7 and later): 512-pair chunks, 256 virtual slices");
#define POCS_LONE_PRE 3        // lone form: iterations of a unit whose normals the block's head draws ahead (72 KB of LDS at 512 threads)
#define POCS_GMM_SUB 32        // units whose wave sums a block holds in LDS at a time (64 runs x 256 slices / 512 blocks)
#define POCS_UNIT_SUMS 10      // survivors + the nine sums
#define POCS_FLUSH_ROWS 5      // flush_unit's transpose scratch per wave: 5 rows of 64 lane values,
#define POCS_FLUSH_PITCH 66    // pitch 66 doubles (16-byte aligned rows, off the bank stride)
// chain record (doubles), one per step i < W-1:
//   [0..2] applied control   [3..5] diag of M   [6..8] the noisy control actually driven (MC)
//   [9] unused               [10 .. 10+L) the L range observations of the step
#define POCS_CHAIN_Z 10
#define POCS_CHAIN_STRIDE 48
static_assert(POCS_CHAIN_Z + POCS_MAX_LANDMARKS <= POCS_CHAIN_STRIDE, "chain record too small");

struct pocs_env_dev {            // collision world as the kernels see it (one copy in HBM, staged to LDS)
  pocs_footprint fp;
  int M;
  int pad;
  double obs[POCS_MAX_OBSTACLES * POCS_OBS_STRIDE];
};

struct pocs_run_header {         // per-run scalars read by every kernel (so a captured graph stays valid)
  uint64_t seed;
  uint64_t pad;                  // sharded whole calls: the context's count of exchange sequences (pocs_gmm_launch::xchg_epoch_from_header)
};

// One-hop exchange of the per-waypoint moments between the GPUs of a node: every rank
// owns a buffer that ALL ranks have mapped (hipIpc); rank q writes its rows into slot q of every
// buffer, then its flags; every rank adds the slots of its own buffer in rank order.
//   data  f64 [2 parities][POCS_XCHG_MAX_WORLD][POCS_XCHG_MAX_RUNS][POCS_XCHG_MAX_NC]
//   flags u64 [2 parities][POCS_XCHG_MAX_WORLD][POCS_XCHG_MAX_RUNS]      = epoch of the row in that slot
#define POCS_XCHG_MAX_WORLD 8
#define POCS_XCHG_MAX_RUNS 256
#define POCS_XCHG_MAX_NC (POCS_MAX_GAUSSIANS * POCS_NMOM)
#define POCS_XCHG_DATA_DOUBLES (2ull * POCS_XCHG_MAX_WORLD * POCS_XCHG_MAX_RUNS * POCS_XCHG_MAX_NC)
#define POCS_XCHG_FLAG_WORDS (2ull * POCS_XCHG_MAX_WORLD * POCS_XCHG_MAX_RUNS)
#define POCS_XCHG_BYTES ((POCS_XCHG_DATA_DOUBLES + POCS_XCHG_FLAG_WORDS + 2) * 8ull)
struct pocs_xchg_dev {
  double* buf[POCS_XCHG_MAX_WORLD];   // buffer of rank q as mapped in THIS process (own rank: the allocation itself)
  int world, rank;
  int parity;                         // which of the two slot sets: (exchanges so far) & 1, the same on every rank
  int pad;
  unsigned long long epoch;           // of this waypoint's rows: (call number << 20) | (waypoint + 1)
};

// Arrays carry a leading "run" dimension: a launch advances `nruns` independent estimations (the
// reference's driver performs 200 of them one after the other, MCSimulation.py:238-256) in lockstep.
struct pocs_gmm_launch {
  const pocs_run_header* hdr;    // [nruns] per-run seeds
  const pocs_env_dev* env;       // obstacle table (records only; M and the footprint travel below)
  const pocs_tables* tables;     // log / sector tables (pocs_math.h), staged to LDS
  const double* chain;           // [nruns][W-1][POCS_CHAIN_STRIDE]
  const pocs_sensor* sensor;
  double* state;                 // [nruns][W][K*POCS_STATE_STRIDE]  mixture sampled at each waypoint
  double* param;                 // [nruns][W][K*POCS_PARAM_STRIDE]  its sampler parameters (incl. the cumulative component counts)
  double* moments;               // [W][nruns][K*POCS_NMOM]          reduced moments of each waypoint
  double* partial;               // [nruns][VS][K*POCS_NMOM]         row of (run, virtual slice); rewritten every waypoint
  const double* partial_prev;    // lone call: the rows the PREVIOUS waypoint's launch left (the other half of the buffer)
  unsigned* sync;                // the call's synchronisation words, zeroed once per call:
                                 // [1] give-up code of a bounded wait (0 = none)
  unsigned* ticket;              // [nruns][W] arrival counters of the blocks of (run, waypoint); same zeroed block
  unsigned* xwait;               // [nruns][W] sharded: how long the closer of (run, waypoint) waited for the world's rows (10 ns ticks); same block
  double* x; double* y; double* th;   // SoA sample buffers [nruns][sample_stride] (unused when !store)
  int16_t* flags;
  long long n_total;             // samples of the whole mixture (all shards): what the component counts add up to
  long long first;               // global index of the shard's first sample
  long long count;               // samples in this shard
  long long sample_stride;       // even, >= count
  int nruns;
  int W;
  pocs_footprint fp;
  double fp_rr, fp_phi;          // its bounding radius and corner angle atan2(hy, hx) (pocs_footprint_extent_pre)
  int M;
  int waypoint;
  int store;
  int advance_in_tail;           // 1: the last block also builds state/param[waypoint+1] (single GPU)
  int exchange_in_tail;          // 1: ... after exchanging the run's moments with the other ranks through `xchg` (sharded)
  int xchg_epoch_from_header;    // 1: the exchange's call number comes from hdr[run].pad (whole calls replayed from a graph), not from xchg
  int lone;                      // 1: one run per call -- no tickets, no closer: every block of waypoint w's launch adds the
                                 //    rows of w-1 and advances the mixture itself, in its head (k_gmm_step, "lone call")
  pocs_xchg_dev xchg;
  // launch geometry (above)
  long long chunks;              // of the shard
  int vs_shift;                  // VS = 1 << vs_shift virtual slices per run
  int upb;                       // units per block (<= VS)
  int blocks;
  int run_lo, run_cnt;           // the runs of the batch this launch works on
};
#define POCS_SYNC_ABORT 1

struct pocs_mc_launch {               // blockIdx.y = run of the batch, like pocs_gmm_launch
  const pocs_run_header* hdr;          // [nruns]
  const pocs_env_dev* env;
  const pocs_tables* tables;
  const double* chain;                 // [nruns][W-1][STRIDE]: noisy control of step s at [..][s][6..8]
  double* x; double* y; double* th;    // SoA particle state [nruns][stride] of this shard
  uint32_t* hits;                      // particlecollisions [nruns][stride]
  unsigned long long* total;           // [nruns] particles with hits > 0
  long long first, count;
  long long stride;                    // >= count
  int W;
  double mu0[3];
  double L0[6];
  int step;                            // k_mc_step: control index; k_mc_fused: number of steps
  int nruns;
  int nontemporal;                     // k_mc_step: the batch's state exceeds the Infinity Cache, stream past it
};


hipError_t pocs_launch_gmm_step(int K, const pocs_gmm_launch& a, hipStream_t s);              // grid = a.blocks
hipError_t pocs_launch_gmm_advance(int K, const pocs_gmm_launch& a, hipStream_t s);
hipError_t pocs_launch_gmm_close(int K, const pocs_gmm_launch& a, hipStream_t s);             // lone call: the last waypoint's rows -> moments
hipError_t pocs_launch_gmm_exchange(int K, const pocs_gmm_launch& a, const pocs_xchg_dev& x, hipStream_t s);   // grid = a.nruns
hipError_t pocs_launch_copy(const void* src, void* dst, long long bytes, hipStream_t s);
hipError_t pocs_launch_fill(void* dst, long long bytes, hipStream_t s);
hipError_t pocs_launch_probe_math(const pocs_tables* tables, int n, const uint32_t* wr, const uint32_t* wa, const double* x, double* out, hipStream_t s);   // out: 5 x n
hipError_t pocs_launch_mc_init(int nblk, const pocs_mc_launch& a, hipStream_t s);
hipError_t pocs_launch_mc_step(int nblk, const pocs_mc_launch& a, hipStream_t s);
hipError_t pocs_launch_mc_fused(int nblk, const pocs_mc_launch& a, hipStream_t s);
hipError_t pocs_launch_mc_count(int nblk, const pocs_mc_launch& a, hipStream_t s);
