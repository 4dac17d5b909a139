// pocs_model.h -- the O(1)-per-waypoint estimator math (product code, host + device).
//
// Each function names the reference function it stands in for (file:line under
// /root/reference/mcsimplugin/).  3x3 matrices are row-major double[9].  Matrix products keep
// the reference's association, (A*B)*C, and a left-to-right inner sum; no operation is
// contracted (the TU is built with -ffp-contract=off and these use no fma on purpose).
// Used by: the host chain (pocs_host.hip) and the per-component device update inside
// mixture advance of k_gmm_step / k_gmm_advance (pocs_kernels.hip).
#pragma once
#include "pocs_math.h"

#define POCS_MAX_LANDMARKS 32
#define POCS_MAX_GAUSSIANS 8
#define POCS_NMOM 11           // per component: nFree nColl Sx Sy St Sxx Sxy Sxt Syy Syt Stt
#define POCS_STATE_STRIDE 16   // per component: mean[3] cov[9] weight alive pad pad
#define POCS_PARAM_STRIDE 12   // per component: mean[3] L00 L10 L11 L20 L21 L22 cumw alive pad

struct pocs_sensor {            // landmarks + sensor variance (MCSimulator.h:97-101)
  double Q;
  int L;
  int pad;
  double lx[POCS_MAX_LANDMARKS];
  double ly[POCS_MAX_LANDMARKS];
};

// angleWrap / roundAngle, MCSimulator.h:56-69: while-loops into [0, 2pi]; exactly 2pi is kept.
// Guard (ours): non-finite or absurd input is returned unchanged instead of looping.
POCS_HD double pocs_wrap_angle(double a) {
  const double TWO_PI = 2 * 3.14159265358979323846;
  if (!(fabs(a) <= 1.0e9)) return a;
  while (a < 0) a += TWO_PI;
  while (a > TWO_PI) a -= TWO_PI;
  return a;
}

// prediction(), MCSimulator.h:413-431 (and moveParticles :300-322, same formula per particle).
POCS_HD void pocs_motion(const double x[3], const double u[3], double out[3]) {
  double sn, cs;
  pocs_sincos(x[2] + u[0], &sn, &cs);
  out[0] = fma(u[1], cs, x[0]);
  out[1] = fma(u[1], sn, x[1]);
  out[2] = pocs_wrap_angle(x[2] + u[0] + u[2]);
}

// EKFpredict(), MCSimulator.h:868-881, with generateG_EKF :517-529, generateV_EKF :453-468
// (V is copied as written: row 2 = [1 0 1]) and M = diag(Md) from generateM_EKF :495-513:
//   R = (V M) V^T,  predSigma = (G Sigma) G^T + R,  predMu = prediction(mu, u).
// G = I + two entries, V has four and M three non-trivial entries; the products below are the
// reference's full 3x3 products with the terms that multiply an exact 0 dropped and the
// multiplications by an exact 1 elided -- the same IEEE values for finite inputs (the oracle keeps
// the full products; tests/test_product_host_vs_oracle.py compares the two bit for bit).
POCS_HD void pocs_ekf_predict(const double mu[3], const double S[9], const double u[3],
                              const double Md[3], double pmu[3], double pS[9]) {
  double sn, cs;
  pocs_sincos(mu[2] + u[0], &sn, &cs);
  const double g02 = -u[1] * sn, g12 = u[1] * cs;      // G(0,2), G(1,2); V(0,0) = g02, V(1,0) = g12
  // GS = G * Sigma
  const double gs00 = S[0] + g02 * S[6], gs01 = S[1] + g02 * S[7], gs02 = S[2] + g02 * S[8];
  const double gs10 = S[3] + g12 * S[6], gs11 = S[4] + g12 * S[7], gs12 = S[5] + g12 * S[8];
  const double gs20 = S[6], gs21 = S[7], gs22 = S[8];
  // T = V * M  (V = [g02 cs 0; g12 sn 0; 1 0 1])
  const double t00 = g02 * Md[0], t01 = cs * Md[1];
  const double t10 = g12 * Md[0], t11 = sn * Md[1];
  const double t20 = Md[0], t22 = Md[2];
  // R = T * V^T
  const double r00 = t00 * g02 + t01 * cs, r01 = t00 * g12 + t01 * sn, r02 = t00;
  const double r10 = t10 * g02 + t11 * cs, r11 = t10 * g12 + t11 * sn, r12 = t10;
  const double r20 = t20 * g02, r21 = t20 * g12, r22 = t20 + t22;
  // predSigma = GS * G^T + R
  pS[0] = (gs00 + gs02 * g02) + r00; pS[1] = (gs01 + gs02 * g12) + r01; pS[2] = gs02 + r02;
  pS[3] = (gs10 + gs12 * g02) + r10; pS[4] = (gs11 + gs12 * g12) + r11; pS[5] = gs12 + r12;
  pS[6] = (gs20 + gs22 * g02) + r20; pS[7] = (gs21 + gs22 * g12) + r21; pS[8] = gs22 + r22;
  pmu[0] = fma(u[1], cs, mu[0]);
  pmu[1] = fma(u[1], sn, mu[1]);
  pmu[2] = pocs_wrap_angle(mu[2] + u[0] + u[2]);
}

// EKFupdate(), MCSimulator.h:883-929 (makeHRow :470-492, observation :368-381): one scalar
// range update per landmark, in landmark order, in place; no angle wrap afterwards.
//   H = [dx/r dy/r 0],  S = (H Sigma) H^T + Q,  K = (Sigma H^T) S^-1,  mu += K (z - r),
//   Sigma = (I - K H) Sigma  -- again with the exact-zero terms of H(2) = 0 dropped.
POCS_HD void pocs_ekf_update(double mu[3], double S[9], const double* z, const pocs_sensor* sen) {
  for (int l = 0; l < sen->L; ++l) {
    const double dx = mu[0] - sen->lx[l];          // == -(lx - mu0), makeHRow's numerator
    const double dy = mu[1] - sen->ly[l];
    const double q = dx * dx + dy * dy;
    const double sq = pocs_sqrt_normal(q);                 // (device: the bare sequences of pocs_math.h; the same values)
    const double rsq = pocs_recip_seed(sq);
    const double H0 = pocs_div_by(dx, sq, rsq);
    const double H1 = pocs_div_by(dy, sq, rsq);
    const double hs0 = H0 * S[0] + H1 * S[3];
    const double hs1 = H0 * S[1] + H1 * S[4];
    const double sinn = (hs0 * H0 + hs1 * H1) + sen->Q;
    const double sinv = pocs_div(1.0, sinn);
    const double K0 = (S[0] * H0 + S[1] * H1) * sinv;
    const double K1 = (S[3] * H0 + S[4] * H1) * sinv;
    const double K2 = (S[6] * H0 + S[7] * H1) * sinv;
    const double innov = z[l] - sq;
    mu[0] = mu[0] + K0 * innov;
    mu[1] = mu[1] + K1 * innov;
    mu[2] = mu[2] + K2 * innov;
    const double a00 = 1.0 - K0 * H0, a01 = 0.0 - K0 * H1;
    const double a10 = 0.0 - K1 * H0, a11 = 1.0 - K1 * H1;
    const double a20 = 0.0 - K2 * H0, a21 = 0.0 - K2 * H1;
    const double n0 = a00 * S[0] + a01 * S[3], n1 = a00 * S[1] + a01 * S[4], n2 = a00 * S[2] + a01 * S[5];
    const double n3 = a10 * S[0] + a11 * S[3], n4 = a10 * S[1] + a11 * S[4], n5 = a10 * S[2] + a11 * S[5];
    const double n6 = (a20 * S[0] + a21 * S[3]) + S[6];
    const double n7 = (a20 * S[1] + a21 * S[4]) + S[7];
    const double n8 = (a20 * S[2] + a21 * S[5]) + S[8];
    S[0] = n0; S[1] = n1; S[2] = n2; S[3] = n3; S[4] = n4; S[5] = n5; S[6] = n6; S[7] = n7; S[8] = n8;
  }
}

// chol(C, "lower") as used by mvnrnd (armadillo_bits/glue_mvnrnd_meat.hpp:92-147 -> LAPACK
// potrf, which reads the lower triangle only).  Returns 0 on a non-positive pivot; the
// reference's eigen-decomposition fallback for that case (:100-132) is not reproduced -- the
// caller retires the component instead (DESIGN.md "degenerate cases").
POCS_HD int pocs_chol3_lower(const double S[9], double L[6]) {
  const double d0 = S[0];
  if (!(d0 > 0.0)) return 0;
  const double l00 = pocs_sqrt_normal(d0);
  const double r00 = pocs_recip_seed(l00);
  const double l10 = pocs_div_by(S[3], l00, r00);
  const double l20 = pocs_div_by(S[6], l00, r00);
  const double d1 = S[4] - l10 * l10;
  if (!(d1 > 0.0)) return 0;
  const double l11 = pocs_sqrt_normal(d1);
  const double l21 = pocs_div(S[7] - l20 * l10, l11);
  const double d2 = (S[8] - l20 * l20) - l21 * l21;
  if (!(d2 > 0.0)) return 0;
  L[0] = l00; L[1] = l10; L[2] = l11; L[3] = l20; L[4] = l21; L[5] = pocs_sqrt_normal(d2);
  return 1;
}

// The per-component tail of truncateGMM(), MCSimulator.h:592-605: mean(free,1) and
// cov(free^T) with Armadillo's single-pass form (op_cov_meat.hpp:26-53):
//   out = A^T A;  out -= acc^T acc / N;  out /= (N-1).
// mom = the 11 accumulated moments of one component.  Returns 0 when fewer than two samples
// survived (the reference has no defined behaviour there).
POCS_HD int pocs_truncated_moments(const double* mom, double mean[3], double cov[9]) {
  const double n = mom[0];
  if (!(n >= 2.0)) return 0;
  const double sx = mom[2], sy = mom[3], st = mom[4];
  const double rn = pocs_recip_seed(n);                   // fifteen quotients, two denominators
  mean[0] = pocs_div_by(sx, n, rn); mean[1] = pocs_div_by(sy, n, rn); mean[2] = pocs_div_by(st, n, rn);
  const double nm1 = n - 1.0;
  const double rm = pocs_recip_seed(nm1);
  const double cxx = pocs_div_by(mom[5] - pocs_div_by(sx * sx, n, rn), nm1, rm);
  const double cxy = pocs_div_by(mom[6] - pocs_div_by(sx * sy, n, rn), nm1, rm);
  const double cxt = pocs_div_by(mom[7] - pocs_div_by(sx * st, n, rn), nm1, rm);
  const double cyy = pocs_div_by(mom[8] - pocs_div_by(sy * sy, n, rn), nm1, rm);
  const double cyt = pocs_div_by(mom[9] - pocs_div_by(sy * st, n, rn), nm1, rm);
  const double ctt = pocs_div_by(mom[10] - pocs_div_by(st * st, n, rn), nm1, rm);
  cov[0] = cxx; cov[1] = cxy; cov[2] = cxt;
  cov[3] = cxy; cov[4] = cyy; cov[5] = cyt;
  cov[6] = cxt; cov[7] = cyt; cov[8] = ctt;
  return 1;
}

// ----------------------------------------------------------------------------------------------
// Component counts of a waypoint: how many of the N samples each component of the mixture gets.
// The reference draws N categorical indices and counts them (GM_Model.h:87-93), i.e. the counts
// are Multinomial(N, weights), and then generates counts[k] samples per component as one block
// (:99-107).  Drawing the counts directly -- K-1 conditional binomials -- gives the same law and
// makes a sample's component a function of its index, so a wave works on one component at a time.
//
// pocs_binomial(n, p): Bin(n, p) for 0 <= n < 2^53 held in a double.  n*min(p,1-p) < 10: waiting
// times (sum of geometric gaps, floor(log U / log q) + 1).  Otherwise BTPE (Kachitvichyanukul &
// Schmeiser, "Binomial random variate generation", CACM 31(2), 1988): triangle / parallelogram /
// exponential-tail envelope, squeeze, Stirling-corrected final test.  Uniforms: 53 bits from two
// words, (m + 1/2) 2^-53 in (0, 1), drawn from Philox stream POCS_STREAM_COUNTS with
// counter = (component, 0, waypoint, stream << 16 | draw number).
// ----------------------------------------------------------------------------------------------
#define POCS_STREAM_COUNTS 4u

struct pocs_count_rng { uint64_t seed; uint32_t comp, waypoint, draw; pocs_u32x4 w; int have; };

POCS_HD double pocs_count_uniform(pocs_count_rng* g) {
  if (g->have == 0) { g->w = pocs_draw(g->seed, (uint64_t)g->comp, g->waypoint, POCS_STREAM_COUNTS, g->draw); g->draw += 1u; g->have = 2; }
  const uint32_t lo = (g->have == 2) ? g->w.x : g->w.z, hi = (g->have == 2) ? g->w.y : g->w.w;
  g->have -= 1;
  const uint64_t m = ((((uint64_t)hi) << 32) | (uint64_t)lo) >> 11;
  return ((double)m + 0.5) * 0x1p-53;
}

POCS_HD double pocs_binomial(double n, double p, pocs_count_rng* g) {
  if (!(n >= 1.0) || !(p > 0.0)) return 0.0;
  if (p >= 1.0) return n;
  const bool flip = p > 0.5;
  const double r = flip ? 1.0 - p : p;
  const double q = 1.0 - r;
  double y;
  if (n * r < 10.0) {
    // waiting times: the successes sit at positions G1, G1+G2, ...; count those <= n
    const double lq = pocs_log(q);
    double pos = 0.0;
    y = 0.0;
    for (int it = 0; it < 400; ++it) {                 // n r < 10: more than 400 successes cannot happen in practice
      pos += floor(pocs_log(pocs_count_uniform(g)) / lq) + 1.0;
      if (pos > n) break;
      y += 1.0;
    }
  } else {
    const double nrq = n * r * q;
    const double fm = n * r + r;
    const double m = floor(fm);
    const double p1 = floor(2.195 * sqrt(nrq) - 4.6 * q) + 0.5;
    const double xm = m + 0.5, xl = xm - p1, xr = xm + p1;
    const double c = 0.134 + 20.5 / (15.3 + m);
    double a = (fm - xl) / (fm - xl * r);
    const double laml = a * (1.0 + a / 2.0);
    a = (xr - fm) / (xr * q);
    const double lamr = a * (1.0 + a / 2.0);
    const double p2 = p1 * (1.0 + 2.0 * c);
    const double p3 = p2 + c / laml;
    const double p4 = p3 + c / lamr;
    y = m;
    for (int it = 0; it < 1000; ++it) {                // acceptance > 0.7 per round: 1000 rounds never exhaust
      const double u = pocs_count_uniform(g) * p4;
      double v = pocs_count_uniform(g);
      if (u <= p1) { y = floor(xm - p1 * v + u); break; }                              // triangle: accept
      if (u <= p2) {                                                                     // parallelograms
        const double x = xl + (u - p1) / c;
        v = v * c + 1.0 - fabs(m - x + 0.5) / p1;
        if (v > 1.0) continue;
        y = floor(x);
      } else if (u <= p3) {                                                              // left tail
        y = floor(xl + pocs_log(v) / laml);
        if (y < 0.0) continue;
        v = v * (u - p2) * laml;
      } else {                                                                           // right tail
        y = floor(xr - pocs_log(v) / lamr);
        if (y > n) continue;
        v = v * (u - p3) * lamr;
      }
      const double k = fabs(y - m);
      if (k > 20.0 && k < nrq / 2.0 - 1.0) {
        // squeeze, then the final test with Stirling's correction
        const double rho = (k / nrq) * ((k * (k / 3.0 + 0.625) + 0.16666666666666666) / nrq + 0.5);
        const double t = -k * k / (2.0 * nrq);
        const double A = pocs_log(v);
        if (A < t - rho) break;
        if (A > t + rho) continue;
        const double x1 = y + 1.0, f1 = m + 1.0, z = n + 1.0 - m, w = n - y + 1.0;
        const double x2 = x1 * x1, f2 = f1 * f1, z2 = z * z, w2 = w * w;
        const double bound = xm * pocs_log(f1 / x1) + (n - m + 0.5) * pocs_log(z / w) + (y - m) * pocs_log(w * r / (x1 * q)) +
                             (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / f2) / f2) / f2) / f2) / f1 / 166320.0 +
                             (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / z2) / z2) / z2) / z2) / z / 166320.0 +
                             (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / x2) / x2) / x2) / x2) / x1 / 166320.0 +
                             (13680.0 - (462.0 - (132.0 - (99.0 - 140.0 / w2) / w2) / w2) / w2) / w / 166320.0;
        if (A > bound) continue;
        break;
      }
      // explicit evaluation of f(y)/f(m) by the recurrence
      const double s = r / q, aa = s * (n + 1.0);
      double F = 1.0;
      if (m < y) { for (double i = m + 1.0; i <= y; i += 1.0) F *= (aa / i - s); }
      else if (m > y) { for (double i = y + 1.0; i <= m; i += 1.0) F /= (aa / i - s); }
      if (v > F) continue;
      break;
    }
  }
  return flip ? n - y : y;
}

// One waypoint of the mixture bookkeeping, split so the device can run one component per
// thread: truncateGMM's tail (:592-629: truncated mean/cov; weights = nFree_k / sum nFree via
// normalise(.,1,1), armadillo_bits/op_normalise_meat.hpp:107-121), then the per-component
// EKFpredict/EKFupdate of EKF_GaussProp (:766-771, :804-812) for the next waypoint, then the
// Cholesky factor and the cumulative weight table the sampler needs.
//   prev   : K x POCS_STATE_STRIDE, the mixture that was sampled at the previous waypoint
//   mom    : K x POCS_NMOM, the (globally reduced) moments of that waypoint; NULL at waypoint 0
//   next   : K x POCS_STATE_STRIDE out, the mixture to sample at this waypoint
//   param  : K x POCS_PARAM_STRIDE out, sampler parameters for this waypoint
// A component with < 2 survivors, or whose covariance is not positive definite, is retired:
// weight 0, alive 0, state frozen.
POCS_HD void pocs_gmm_advance_component(int k, const double* prev, const double* mom,
                                        const double* u, const double* Md, const double* z,
                                        const pocs_sensor* sen, double* next, double* param) {
  const double* p = prev + k * POCS_STATE_STRIDE;
  double* o = next + k * POCS_STATE_STRIDE;
  double mean[3], cov[9];
  double alive = p[13];
  double weight = p[12];
  for (int i = 0; i < 3; ++i) mean[i] = p[i];
  for (int i = 0; i < 9; ++i) cov[i] = p[3 + i];
  if (mom) {
    weight = 0.0;
    if (alive != 0.0) {
      double tm[3], tc[9];
      if (pocs_truncated_moments(mom + k * POCS_NMOM, tm, tc)) {
        double pm[3], pc[9];
        pocs_ekf_predict(tm, tc, u, Md, pm, pc);
        pocs_ekf_update(pm, pc, z, sen);
        for (int i = 0; i < 3; ++i) mean[i] = pm[i];
        for (int i = 0; i < 9; ++i) cov[i] = pc[i];
        weight = mom[k * POCS_NMOM];       // nFree_k, normalised by pocs_gmm_normalise
      } else {
        alive = 0.0;
      }
    }
  }
  double L[6] = {0, 0, 0, 0, 0, 0};
  if (alive != 0.0 && !pocs_chol3_lower(cov, L)) { alive = 0.0; weight = 0.0; }
  for (int i = 0; i < 3; ++i) o[i] = mean[i];
  for (int i = 0; i < 9; ++i) o[3 + i] = cov[i];
  o[12] = weight; o[13] = alive; o[14] = 0.0; o[15] = 0.0;
  double* q = param + k * POCS_PARAM_STRIDE;
  for (int i = 0; i < 3; ++i) q[i] = mean[i];
  for (int i = 0; i < 6; ++i) q[3 + i] = L[i];
  q[9] = 0.0; q[10] = alive; q[11] = 0.0;
}

// Second half: weights and the sampler's selection table.  `renorm` = 1 when the weights in
// `next` are raw survivor counts (every waypoint but the first).  The table is the cumulative
// COUNT of samples per component for this waypoint, param[k][9] = n_0 + ... + n_k with
// (n_0 .. n_{K-1}) ~ Multinomial(n_total, weights) drawn as conditional binomials in component
// order (a retired component has weight 0 and gets none; if every component is retired,
// component 0's frozen mean receives all samples): global sample i belongs to the first
// component whose entry exceeds i.
POCS_HD int pocs_normalise_weights(int K, int renorm, double* next) {
  double wsum = 0.0;
  for (int k = 0; k < K; ++k) wsum += next[k * POCS_STATE_STRIDE + 12];
  // normalise(collisionCounts,1,1).row(1): divide by the L1 norm, a zero norm divides by 1.
  const double den = (wsum != 0.0) ? wsum : 1.0;
  int last_alive = -1;
  for (int k = 0; k < K; ++k) {
    double* o = next + k * POCS_STATE_STRIDE;
    if (renorm) o[12] = o[12] / den;
    if (o[13] != 0.0 && o[12] > 0.0) last_alive = k;
  }
  return last_alive;
}

// Cumulative component counts from normalised weights (next[k][12]) and alive flags (next[k][13]).
POCS_HD void pocs_component_counts(int K, const double* next, int last_alive, uint64_t seed,
                                   uint32_t waypoint, double n_total, double* cum, int cum_stride) {
  double suffix[POCS_MAX_GAUSSIANS];
  double tail = 0.0;
  for (int k = K - 1; k >= 0; --k) {
    const double* o = next + k * POCS_STATE_STRIDE;
    if (o[13] != 0.0 && o[12] > 0.0) tail += o[12];
    suffix[k] = tail;
  }
  double rem = n_total, run = 0.0;
  for (int k = 0; k < K; ++k) {
    const double* o = next + k * POCS_STATE_STRIDE;
    double nk = 0.0;
    if (last_alive < 0) {
      nk = (k == 0) ? rem : 0.0;
    } else if (k == last_alive) {
      nk = rem;
    } else if (k < last_alive && o[13] != 0.0 && o[12] > 0.0) {
      pocs_count_rng g; g.seed = seed; g.comp = (uint32_t)k; g.waypoint = waypoint; g.draw = 0u; g.have = 0;
      nk = pocs_binomial(rem, o[12] / suffix[k], &g);
    }
    rem -= nk;
    run += nk;
    cum[k * cum_stride] = run;
  }
}

POCS_HD void pocs_gmm_normalise(int K, int renorm, double* next, double* param, uint64_t seed,
                                uint32_t waypoint, double n_total) {
  const int last_alive = pocs_normalise_weights(K, renorm, next);
  pocs_component_counts(K, next, last_alive, seed, waypoint, n_total, param + 9, POCS_PARAM_STRIDE);
}
