// pocs_tuning.h -- diagnostic hooks of pocs_kernels.hip / pocs_host.hip.  NOT part of the shipped library:
// included only when the translation units are compiled with -DPOCS_TUNING (tools/ablate.sh, tools/jobs/stamps.sh);
// the default build defines the hooks below as no-ops / pass-throughs and sees none of this file.
//
//   -DPOCS_TUNING -DPOCS_STAMPS           per-block phase times of k_gmm_step (head, units, barrier, rows + drain,
//                                         ticket; closers: row sums, advance and its pieces) added up in device
//                                         counters that pocs_destroy prints (pocs_stamps_report)
//   -DPOCS_TUNING -DPOCS_ABLATE_<PART>    timing-only builds with one part of the sampling body replaced by something
//                                         trivial (RNG, PHILOX, BOXMULLER, COLLIDE, MOMENTS; combinable): WRONG outputs,
//                                         same launch structure -- what the part costs in situ
//   -DPOCS_TUNING -DPOCS_CALL_TIMES       host-side phases of every whole-run GMM call on stderr
//
// Retired in round 4 (losers of rounds 2 and 3, in the history): priorities rotated by time, unequal shares for a
// CU's two blocks, the literal-constant form of the polynomials, block-size / blocks-per-CU / slice-count sweeps.
#ifndef POCS_TUNING_HOOKS
#define POCS_TUNING_HOOKS

#if defined(POCS_STAMPS)
__device__ unsigned long long g_stamps[24 * 256];            // a set per block (modulo 256): blocks in lockstep must not queue on one word
#define POCS_STAMP_AT(i) (&g_stamps[(i) + 24 * (blockIdx.x & 255)])
#define POCS_STAMP_BEGIN() unsigned long long last_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd(POCS_STAMP_AT(15), 1ull)
#define POCS_STAMP(i) do { if (threadIdx.x == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(POCS_STAMP_AT(i), n_ - last_); last_ = n_; } } while (0)
#define POCS_STAMP_COUNT(i) do { if (threadIdx.x == 0) atomicAdd(POCS_STAMP_AT(i), 1ull); } while (0)
#define POCS_ADV_STAMP_BEGIN() unsigned long long t_ = wall_clock64()
#define POCS_ADV_STAMP(i) do { if (tid == 0) { const unsigned long long n_ = wall_clock64(); atomicAdd(POCS_STAMP_AT(i), n_ - t_); t_ = n_; } } while (0)
// how often the footprint test has anything to do: iterations (a wave's 128 samples), those of runs that kept any obstacle
// record, those with ANY lane inside any kept record's broad-phase box; kept records and such lanes per iteration
#define POCS_TUNE_COLLIDE_STATS() do { bool pass_ = false; \
    for (int m_ = 0; m_ < nkeep; ++m_) { const double* o_ = s_keep + m_ * POCS_OBS_STRIDE; \
      for (int h_ = 0; h_ < 2; ++h_) pass_ = pass_ || (fabs(o_[0] - xs[h_]) <= o_[6] && fabs(o_[1] - ys[h_]) <= o_[7]); } \
    const unsigned long long b_ = __ballot(pass_); \
    bool own_ = false; { const double* p_ = &s_par[ks[0] * POCS_PARAM_STRIDE]; const double ex_ = 6.67 * fabs(p_[3]), ey_ = 6.67 * (fabs(p_[4]) + fabs(p_[5])); \
      for (int m_ = 0; m_ < nkeep; ++m_) { const double* o_ = s_keep + m_ * POCS_OBS_STRIDE; \
        own_ = own_ || (fabs(o_[0] - p_[0]) <= o_[6] + ex_ && fabs(o_[1] - p_[1]) <= o_[7] + ey_); } } \
    const unsigned long long c_ = __ballot(own_); \
    if ((threadIdx.x & 63) == 0) { atomicAdd(POCS_STAMP_AT(16), 1ull); if (nkeep > 0) atomicAdd(POCS_STAMP_AT(17), 1ull); if (b_) atomicAdd(POCS_STAMP_AT(18), 1ull); \
      atomicAdd(POCS_STAMP_AT(19), (unsigned long long)nkeep); atomicAdd(POCS_STAMP_AT(20), (unsigned long long)__popcll(b_)); if (c_ & 1ull) atomicAdd(POCS_STAMP_AT(21), 1ull); } } while (0)
#else
#define POCS_TUNE_COLLIDE_STATS() do { } while (0)
#define POCS_STAMP_BEGIN() do { } while (0)
#define POCS_STAMP(i) do { } while (0)
#define POCS_STAMP_COUNT(i) do { } while (0)
#define POCS_ADV_STAMP_BEGIN() do { } while (0)
#define POCS_ADV_STAMP(i) do { } while (0)
#endif

// The sampling body's three parts as the iteration of gmm_units names them (zz, spare, lp, pair0, seed, w, s_tab, xs,
// ys, ts, hits, two, ks, acc, nfree are the iteration's locals).
#if defined(POCS_ABLATE_RNG)          // no Philox, no Box-Muller
#define POCS_TUNE_NORMALS(...) for (int h = 0; h < 2; ++h) { zz[h][0] = (double)(lp & 7) * 0.1; zz[h][1] = (double)(lp & 3) * 0.1; zz[h][2] = 0.05; spare[h] = (uint32_t)lp * 2654435761u; }
#elif defined(POCS_ABLATE_PHILOX)     // Box-Muller kept, its words from a three-instruction hash: what Philox costs in situ
#define POCS_TUNE_NORMALS(...) { const uint32_t q = (uint32_t)(pair0 + lp) * 2654435761u ^ (uint32_t)seed ^ ((uint32_t)w << 20); \
      const uint32_t a0 = q * 0x9E3779B1u, a1 = (q ^ 0x85EBCA6Bu) * 0xC2B2AE35u, a2 = (q + 0x27D4EB2Fu) * 0x165667B1u; \
      pocs_normal_pair_w2(a0, a1, s_tab, &zz[0][0], &zz[0][1]); \
      pocs_normal_pair_w2(a2, a0 ^ a1, s_tab, &zz[0][2], &zz[1][0]); \
      pocs_normal_pair_w2(a1 ^ a2, a0 + a2, s_tab, &zz[1][1], &zz[1][2]); \
      spare[0] = a0; spare[1] = a1; }
#elif defined(POCS_ABLATE_BOXMULLER)  // Philox kept, the normals a scaling of its words
#define POCS_TUNE_NORMALS(...) { const pocs_u32x4 A = pocs_draw(seed, pair0 + lp, (uint32_t)w, POCS_STREAM_GMM, 0u), B = pocs_draw(seed, pair0 + lp, (uint32_t)w, POCS_STREAM_GMM, 1u); \
      zz[0][0] = (double)A.x * 0x1p-32; zz[0][1] = (double)A.y * 0x1p-32; zz[0][2] = (double)A.z * 0x1p-32; spare[0] = B.z; \
      zz[1][0] = (double)A.w * 0x1p-32; zz[1][1] = (double)B.x * 0x1p-32; zz[1][2] = (double)B.y * 0x1p-32; spare[1] = B.w; }
#else
#define POCS_TUNE_NORMALS(...) __VA_ARGS__
#endif
#if defined(POCS_ABLATE_COLLIDE)
#define POCS_TUNE_COLLIDE(...) hits[0] = xs[0] > ts[0]; hits[1] = xs[1] > ts[1]
#else
#define POCS_TUNE_COLLIDE(...) __VA_ARGS__
#endif
#if defined(POCS_ABLATE_MOMENTS)
#define POCS_TUNE_SKIP_MOMENTS true
#define POCS_TUNE_MOMENTS_ALT() do { acc[1] += xs[0] + ys[0] + ts[0] + xs[1]; nfree += (hits[0] || (two && ks[1] == 0)) ? 0 : 1; } while (0)
#else
#define POCS_TUNE_SKIP_MOMENTS false
#define POCS_TUNE_MOMENTS_ALT() do { } while (0)
#endif

#endif  // POCS_TUNING_HOOKS

#if defined(POCS_STAMPS) && defined(POCS_TUNING_REPORT)      // (defined by pocs_kernels.hip, once, behind its kernels)
extern "C" void pocs_stamps_report() {
  static unsigned long long all[24 * 256];
  unsigned long long h[24] = {0};
  if (hipMemcpyFromSymbol(all, HIP_SYMBOL(g_stamps), sizeof all) != hipSuccess) return;
  for (int b = 0; b < 256; ++b) for (int i = 0; i < 24; ++i) h[i] += all[24 * b + i];
  if (h[15] == 0) return;
  const double nb = (double)h[15], nc = (double)(h[14] ? h[14] : 1);
  fprintf(stderr, "[stamps] %.0f blocks, %.0f closers; per block (us): head %.2f | units %.2f | -> barrier %.2f | rows + drain + barrier %.2f | "
          "ticket + barrier %.2f ; per closer: close_sums %.2f | advance %.2f (staging %.2f, components (wave 0) %.2f, -> the counts lane %.2f, normalise + publish + drain %.2f)\n",
          nb, nc, 0.01 * h[0] / nb, 0.01 * h[1] / nb, 0.01 * h[2] / nb, 0.01 * h[3] / nb, 0.01 * h[4] / nb, 0.01 * h[5] / nc, 0.01 * h[6] / nc,
          0.01 * h[8] / nc, 0.01 * h[9] / nc, 0.01 * h[10] / nc, 0.01 * h[11] / nc);
  if (h[16]) fprintf(stderr, "[stamps] footprint test: %llu wave iterations, %.1f %% in runs with kept records (%.2f records per iteration), %.1f %% with a lane inside a record's broad-phase box (%.2f lanes per iteration); %.1f %% where the box of the wave's OWN component (first lane's) reaches a kept record\n",
                     h[16], 100.0 * h[17] / h[16], (double)h[19] / h[16], 100.0 * h[18] / h[16], (double)h[20] / h[16], 100.0 * h[21] / h[16]);
  for (auto& v : all) v = 0;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), all, sizeof all);
}
#endif
